"""CPU-side checks of the C-ABI library and host logic (no GPU, no compute calls)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from iodine_amd import _lib, synth
from oracle import iodine_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    L = _lib.lib()
    header = open(os.path.join(ROOT, 'include', 'iodine_hip.h')).read()
    declared = set(re.findall(r'\b(iodine_[a-z0-9_]+)\s*\(', header))
    declared -= {'iodine_handle', 'iodine_config'}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.iodine_abi_version() == 3


def test_header_documents_every_option_and_profile_category():
    """include/iodine_hip.h is the contract: every key iodine_set_option accepts, every category a launch is profiled under and the
    `seen:` query must be named in it (five options and two categories had been added to the library only)."""
    header = open(os.path.join(ROOT, 'include', 'iodine_hip.h')).read()
    api = open(os.path.join(ROOT, 'iodine_amd', 'csrc', 'iodine_api.cpp')).read()
    body = api[api.index('int iodine_set_option('):api.index('int iodine_reconstruct(')]
    options = set(re.findall(r'strcmp\(key, "([a-z_0-9]+)"\)', body))
    assert {'conv_precision', 'refine_l0_fused', 'profile_stride'} <= options            # the scan itself works
    missing = sorted(o for o in options if f'"{o}"' not in header)
    assert not missing, f'options accepted by iodine_set_option but absent from the header: {missing}'
    cats = set(re.findall(r'PROF\(h, st, (?:[^",;]*\? *)?"([a-z_0-9]+)"', api)) | set(re.findall(r'\? "([a-z_0-9]+)" : "([a-z_0-9]+)"', api)[0])
    assert {'conv_tile_fwd', 'refine_l0f', 'refine_bwd01', 'dec_out_bwd'} <= cats
    missing = sorted(c for c in cats if f'"{c}"' not in header)
    assert not missing, f'profile categories absent from the header: {missing}'
    assert '"seen:<category>"' in header


def test_config_struct_matches_header_field_order():
    header = open(os.path.join(ROOT, 'include', 'iodine_hip.h')).read()
    body = header[header.index('typedef struct iodine_config {'):header.index('} iodine_config;')]
    fields = re.findall(r'^\s*(?:int|double|unsigned)\s+([a-z_]+);', body, flags=re.M)
    assert fields == [f[0] for f in _lib.Config._fields_]


def test_create_rejects_unsupported_configs_with_message():
    L = _lib.lib()
    cfg = _lib.Config(dim_latent=16, iters=3, slots=4, img_size=64, img_channels=3, sigma=0.1, layernorm=1,
                      stop_gradient=0, encoding=_lib.ENC_FULL & ~(1 << 1), ref_conv_chan=32, ref_conv_layers=3,
                      ref_mlp_units=128, ref_kernel_size=3, ref_stride=2, dec_conv_chan=32, dec_conv_layers=5,
                      dec_kernel_size=3)
    h = C.c_void_p()
    assert L.iodine_create(C.byref(cfg), C.byref(h)) == 1 and not h.value      # no 'grad_post': the LSTM input would be 2L narrower
    assert b'ENCODING' in L.iodine_last_error(None) and b'grad_post' in L.iodine_last_error(None)
    cfg.encoding = 0x3                                                          # only the latent entries: no conv input at all
    assert L.iodine_create(C.byref(cfg), C.byref(h)) == 1 and b'image-shaped' in L.iodine_last_error(None)
    cfg.encoding = _lib.ENC_FULL
    cfg.dec_kernel_size = 4                       # even kernel sizes do not exist in the reference (padding = KERNEL_SIZE // 2)
    assert L.iodine_create(C.byref(cfg), C.byref(h)) == 1
    assert b'KERNEL_SIZE' in L.iodine_last_error(None)
    cfg.dec_kernel_size = 5
    cfg.ref_conv_chan = 48                        # the refinement head pools in groups that divide 256
    assert L.iodine_create(C.byref(cfg), C.byref(h)) == 1
    assert b'REF.CONV_CHAN' in L.iodine_last_error(None)


def test_linspace_matches_torch():
    L = _lib.lib()
    for n in (16, 64, 128):
        out = (C.c_float * n)()
        L.iodine_linspace_host(n, out)
        ref = torch.linspace(-1, 1, n).numpy()
        got = np.frombuffer(out, dtype=np.float32)
        assert np.abs(got - ref).max() <= 1.2e-7         # <= 1 ulp (ATen's vectorised path rounds differently)
        assert got[0] == -1.0 and got[-1] == 1.0


def test_module_surface_matches_reference_names():
    from util import golden_setup, hip_arch, load_golden
    from iodine_amd import IODINE
    g = load_golden('tiny')
    arch, params, _, _, _ = golden_setup(g)
    m = IODINE(hip_arch(arch))
    assert [(k, tuple(v.shape)) for k, v in m.named_parameters()] == [(k, tuple(v.shape)) for k, v in params.items()]
    assert m.get_input_size() == (17, 4 * arch.dim_latent)
    assert m.sigma == arch.sigma and m.K == arch.slots and m.n_iters == arch.iters
    with pytest.raises(RuntimeError, match='ROCm device'):
        m.reconstruct(torch.zeros(1, 3, arch.img_size, arch.img_size))


def test_max_batch_and_chunking_plan():
    """IODINE.max_batch mirrors the library's 32-bit offset limits (iodine_api.cpp check_ready / iodine_train_forward); larger
    batches are cut into balanced runs of independent images."""
    from iodine_amd import IODINE
    from iodine_amd.model import clevr6_arch
    m = IODINE(clevr6_arch())                                   # cfg3: K = 7, T = 5, 128 x 128, 64 channels
    assert m.max_batch() == (2 ** 31 - 1) // (7 * 128 * 128 * 64) == 292
    assert m.max_batch(training=True) == (2 ** 31 - 1) // (7 * 5 * 128 * 128 * 20) == 187
    m.set_option('batch_cap', 32)
    assert m.max_batch() == m.max_batch(training=True) == 32
    assert m._chunks(256, 32) == [(32 * i, 32 * i + 32) for i in range(8)]
    assert m._chunks(70, 32) == [(0, 24), (24, 47), (47, 70)]
    assert m._chunks(5, 32) == [(0, 5)]


def test_synth_is_deterministic_and_shard_invariant():
    e1 = synth.make_eps(2, 4, 3, 8, seed=1)
    e2 = synth.make_eps(2, 4, 3, 8, seed=1)
    assert np.array_equal(e1, e2)
    assert abs(float(e1.mean())) < 0.2 and 0.8 < float(e1.std()) < 1.2
    a = synth.make_images(4, 16, seed=0)
    b = synth.make_images(2, 16, seed=0, first_index=2)
    assert np.array_equal(a[2:], b)
    imgs, masks = synth.make_images(2, 32, seed=0, kind='blobs')
    assert imgs.min() >= 0 and imgs.max() <= 1 and len(masks) == 2 and masks[0].shape[1:] == (32, 32)


def test_reference_checkpoint_loads_on_cpu_side(tmp_path):
    """A file written the way lib/utils/checkpoint.py:36-54 writes it (DataParallel 'module.' prefix) loads into the
    module's parameter table; no GPU involved (load_state_dict only)."""
    from util import golden_setup, hip_arch, load_golden
    from iodine_amd import IODINE, checkpoint
    g = load_golden('tiny')
    arch, params, _, _, _ = golden_setup(g)
    path = str(tmp_path / 'ref.pth')
    torch.save({'model': {'module.' + k: v for k, v in params.items()}, 'optimizer': None, 'epoch': 3}, path)
    m = IODINE(hip_arch(arch))
    extra = checkpoint.load_checkpoint(path, m)
    assert extra == {'epoch': 3}
    for k, v in m.state_dict().items():
        assert torch.equal(v, params[k]), k


def test_ablation_hook_build_of_the_host_file_compiles():
    """tools/helper_cost.py and the DESIGN.md helper-cost figures come from a -DIODINE_XSKIP_HOOK build; the guard sits in front of
    an if / else in decoder_backward_data and once swallowed a declaration: keep that build compiling."""
    import shutil
    import subprocess
    hipcc = '/opt/rocm/bin/hipcc' if os.path.exists('/opt/rocm/bin/hipcc') else shutil.which('hipcc')
    if not hipcc:
        pytest.skip('hipcc not installed')
    src = os.path.join(ROOT, 'iodine_amd', 'csrc', 'iodine_api.cpp')
    r = subprocess.run([hipcc, '-fsyntax-only', '--cuda-host-only', '--offload-arch=gfx950', '-std=c++17', '-x', 'hip', '-Wno-dangling-else',
                        '-Wno-unused-command-line-argument', '-DIODINE_XSKIP_HOOK', src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
