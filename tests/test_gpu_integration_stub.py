"""INTEGRATION.md section 2 performed verbatim: raw ``ctypes`` against libiodine_hip.so, no iodine_amd.model / _lib - the calls a
maintainer of the reference would write around lib/engine/train.py:60-63.  Gradients go through ``iodine_train_backward``
(POINTER-ARRAY form) into separately allocated ``.grad`` tensors, twice with grad_scale 0.5 (``+=`` accumulation like autograd's
.grad), and are compared with the reference's own numbers (tests/golden/tiny.npz)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from util import golden_setup, load_golden, rel_l2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Config(C.Structure):                    # mirrors ARCH.* (iodine.py:8-32), field order of include/iodine_hip.h
    _fields_ = [('dim_latent', C.c_int), ('iters', C.c_int), ('slots', C.c_int), ('img_size', C.c_int),
                ('img_channels', C.c_int), ('sigma', C.c_double), ('layernorm', C.c_int), ('stop_gradient', C.c_int),
                ('encoding', C.c_uint), ('ref_conv_chan', C.c_int), ('ref_conv_layers', C.c_int),
                ('ref_mlp_units', C.c_int), ('ref_kernel_size', C.c_int), ('ref_stride', C.c_int),
                ('dec_conv_chan', C.c_int), ('dec_conv_layers', C.c_int), ('dec_kernel_size', C.c_int)]


def test_integration_stub_pointer_array_backward():
    lib = C.CDLL(os.path.join(ROOT, 'iodine_amd', 'libiodine_hip.so'))
    lib.iodine_last_error.restype = C.c_char_p
    lib.iodine_last_error.argtypes = [C.c_void_p]
    lib.iodine_workspace_bytes.restype = C.c_size_t
    lib.iodine_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int]
    vp = C.c_void_p
    lib.iodine_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.iodine_destroy.argtypes = [vp]
    lib.iodine_destroy.restype = None
    lib.iodine_set_params.argtypes = [vp, vp, C.POINTER(vp), C.c_int]
    lib.iodine_set_workspace.argtypes = [vp, vp, C.c_size_t]
    lib.iodine_train_forward.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp]
    lib.iodine_train_backward.argtypes = [vp, vp, C.c_float, C.POINTER(vp), C.c_int]

    g = load_golden('tiny')
    arch, params, x, eps, _ = golden_setup(g)
    dev = 'cuda:0'
    B = x.shape[0]
    h = vp()
    cfg = Config(arch.dim_latent, arch.iters, arch.slots, arch.img_size, 3, arch.sigma, int(arch.layernorm), 0, 0xFFF,
                 arch.ref_chan, arch.ref_layers, arch.ref_mlp, 3, 2, arch.dec_chan, arch.dec_layers, 3)
    assert lib.iodine_create(C.byref(cfg), C.byref(h)) == 0, lib.iodine_last_error(None)
    try:
        # "reference module on the GPU": its parameters in named_parameters() order, each with its own .grad tensor
        ps = [v.to(dev).contiguous() for v in params.values()]
        grads = [torch.zeros_like(p) for p in ps]                                  # optimizer.zero_grad(): train.py:62
        ptrs = (vp * len(ps))(*[p.data_ptr() for p in ps])
        stream = vp(torch.cuda.current_stream().cuda_stream)
        assert lib.iodine_set_params(h, stream, ptrs, len(ps)) == 0, lib.iodine_last_error(h)
        ws = torch.empty(lib.iodine_workspace_bytes(h, B, 1), dtype=torch.uint8, device=dev)
        assert lib.iodine_set_workspace(h, vp(ws.data_ptr()), ws.numel()) == 0, lib.iodine_last_error(h)
        xd, ed = x.to(dev).contiguous(), eps.to(dev).contiguous()
        loss = torch.empty((), device=dev)
        elbo_iter = torch.empty((arch.iters + 1, 3), device=dev)
        gp = (vp * len(ps))(*[t.data_ptr() for t in grads])
        for _ in range(2):                                                         # two half-weighted backward passes accumulate
            assert lib.iodine_train_forward(h, stream, B, xd.data_ptr(), ed.data_ptr(), loss.data_ptr(), elbo_iter.data_ptr()) == 0, \
                lib.iodine_last_error(h)
            assert lib.iodine_train_backward(h, stream, C.c_float(0.5), gp, len(ps)) == 0, lib.iodine_last_error(h)
        # a second backward of the same forward is refused (autograd without retain_graph)
        assert lib.iodine_train_backward(h, stream, C.c_float(0.5), gp, len(ps)) == 3
        torch.cuda.synchronize()
        ref_loss = float(g['f32.train.loss'])
        assert abs(loss.item() - ref_loss) <= 1e-5 * abs(ref_loss)
        assert np.abs(elbo_iter[:, 0].cpu().numpy() - g['f32.train.elbos']).max() <= 1e-5 * np.abs(g['f32.train.elbos']).max()
        bad = [(n, rel_l2(t.cpu().numpy(), g['f32.train.grad.' + n])) for n, t in zip(params.keys(), grads)
               if not rel_l2(t.cpu().numpy(), g['f32.train.grad.' + n]) < 2e-4]
        assert not bad, bad
        # a NULL entry skips that parameter; the others still accumulate (now 1.5x)
        gp[0] = None
        before = grads[0].clone()
        assert lib.iodine_train_forward(h, stream, B, xd.data_ptr(), ed.data_ptr(), loss.data_ptr(), elbo_iter.data_ptr()) == 0
        assert lib.iodine_train_backward(h, stream, C.c_float(0.5), gp, len(ps)) == 0, lib.iodine_last_error(h)
        torch.cuda.synchronize()
        assert torch.equal(grads[0], before)
        n1 = list(params.keys())[1]
        assert rel_l2(grads[1].cpu().numpy() / 1.5, g['f32.train.grad.' + n1]) < 2e-4
    finally:
        lib.iodine_destroy(h)
