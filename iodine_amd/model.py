"""``IODINE(ARCH)``: the reference's module surface over libiodine_hip.so.

Mirrors ``lib/modeling/iodine.py`` of zhixuan-lin/IODINE for the callers of the hot path:
``loss = model(x)`` (lib/engine/train.py:60), ``model.reconstruct(image)``
(lib/eval/ari_eval.py:22), ``encode`` / ``decode`` (iodine.py:59-112), the
``named_parameters()`` / ``state_dict()`` names (lib/solver/build.py:10-14,
lib/utils/checkpoint.py:43,68), ``model.sigma`` (train.py:97) and the ``logger`` side
channel keys (iodine.py:156-157,226-239).  All arithmetic runs in hand-written gfx950
kernels behind the C ABI of include/iodine_hip.h; PyTorch only owns the tensors, the
stream and the autograd hook-up.  There is no CPU or eager fallback: tensors must live on
a ROCm device and the shared library must be built.

Extensions over the reference: every entry point takes an optional ``eps`` tensor of shape
(T+1, B, K, L) replacing the ``torch.randn_like`` draws of ``Gaussian.sample``
(iodine.py:632) in call order, so that results can be compared with the CPU oracle.  Without
``eps`` the draws come from the library's own counter-based generator (Philox4x32-10,
``iodine_randn``; seed with ``model.manual_seed``) - no ATen COMPUTE kernel runs on the product path (what a kernel trace
still shows from ATen are autograd's own fills: ``loss.backward()`` seeds the scalar loss gradient with ``ones_like`` and
``zero_grad`` / the flat gradient buffer are fills - a handful of 5-us launches per training step, none of them arithmetic of the model).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch
from torch import nn

from . import _lib


class Logger:
    """Same role as lib/utils/vis_logger.py:30-50: a dict of the latest values."""

    def __init__(self):
        self.things = dict()

    def __getitem__(self, key):
        return self.things[key]

    def __contains__(self, key):
        return key in self.things

    def update(self, **kw):
        self.things.update(kw)


logger = Logger()


def _arch_get(arch, name, default=None):
    return getattr(arch, name, default)


class _MultiLayerConv(nn.Module):
    """Parameter container with the reference's names (iodine.py:570-584); never called."""

    def __init__(self, dim_in, dim_out, n_layers, kernel_size, stride=1):
        super().__init__()
        self.layers = nn.ModuleList()
        for _ in range(n_layers):
            self.layers.append(nn.Conv2d(dim_in, dim_out, kernel_size, stride=stride, padding=kernel_size // 2))
            dim_in = dim_out


class _MLP(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.layers = nn.ModuleList([nn.Linear(dim_in, dim_out)])       # iodine.py:553-557


class _Refine(nn.Module):
    def __init__(self, dim_in, dim_conv, dim_hidden, dim_out, n_layers, kernel_size, stride):
        super().__init__()                                              # iodine.py:450-464
        self.mlc = _MultiLayerConv(dim_in, dim_conv, n_layers, kernel_size, stride)
        self.mlp = _MLP(dim_conv, dim_hidden)
        self.lstm = nn.LSTMCell(dim_hidden + 4 * dim_out, dim_hidden)
        self.mean_update = nn.Linear(dim_hidden, dim_out)
        self.logvar_update = nn.Linear(dim_hidden, dim_out)


class _Decoder(nn.Module):
    def __init__(self, dim_in, dim_hidden, n_layers, kernel_size):
        super().__init__()                                              # iodine.py:416-423
        self.mlc = _MultiLayerConv(dim_in + 2, dim_hidden, n_layers, kernel_size)
        self.conv = nn.Conv2d(dim_hidden, 4, kernel_size, stride=1, padding=kernel_size // 2)


class _Posterior(nn.Module):
    def __init__(self, dim_latent):
        super().__init__()                                              # iodine.py:596-604
        self.init_mean = nn.Parameter(torch.zeros(dim_latent))
        self.init_logvar = nn.Parameter(torch.zeros(dim_latent))
        self.mean = None
        self.logvar = None


class _TrainStep(torch.autograd.Function):
    """``loss = model(x)`` / ``loss.backward()`` through iodine_train_forward / iodine_train_backward."""

    @staticmethod
    def forward(ctx, module, x, eps, *params):
        loss, elbo_iter = module._train_forward(x, eps)
        ctx.module = module
        ctx.serial = module._call_serial            # identity of the saved forward (the library keeps exactly one)
        ctx.n = len(params)
        ctx.mark_non_differentiable(elbo_iter)
        return loss, elbo_iter

    @staticmethod
    def backward(ctx, grad_loss, _grad_elbo):
        grads = ctx.module._train_backward(grad_loss, ctx.serial)
        return (None, None, None, *grads)


_WRAPPER_OPTIONS = ('batch_cap',)      # max images per library call (IODINE.max_batch); the rest go to iodine_set_option


class _ChunkedTrainStep(torch.autograd.Function):
    """The same for a batch larger than one device call takes (``IODINE.max_batch``): the images are independent and the loss is
    a batch mean, so the batch runs as chunks, each chunk's forward AND backward at once (the library keeps one saved forward),
    the parameter gradients accumulated with the chunk's share of the batch; ``loss.backward()`` then only scales them."""

    @staticmethod
    def forward(ctx, module, x, eps, *params):
        loss, elbo_iter, flat = module._train_chunked(x, eps)
        ctx.module, ctx.flat = module, flat
        ctx.mark_non_differentiable(elbo_iter)
        return loss, elbo_iter

    @staticmethod
    def backward(ctx, grad_loss, _grad_elbo):
        flat = ctx.flat * grad_loss.to(ctx.flat.dtype)
        views, off = [], 0
        for p in ctx.module._ordered_params():
            views.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        return (None, None, None, *views)


class IODINE(nn.Module):
    def __init__(self, ARCH):
        super().__init__()
        # same attribute reads as iodine.py:8-21
        self.dim_latent = ARCH.DIM_LATENT
        self.n_iters = ARCH.ITERS
        self.K = ARCH.SLOTS
        self.encodings = list(ARCH.ENCODING)
        self.img_channels = ARCH.IMG_CHANNELS
        self.img_size = ARCH.IMG_SIZE
        self.sigma = ARCH.SIGMA
        self.use_layernorm = ARCH.LAYERNORM
        self.use_stop_gradient = _arch_get(ARCH, 'STOP_GRADIENT', False)

        input_size, lambda_size = self.get_input_size()
        ref, dec = ARCH.REF, ARCH.DEC
        self.refine = _Refine(input_size, ref.CONV_CHAN, ref.MLP_UNITS, ARCH.DIM_LATENT, ref.CONV_LAYERS,
                              ref.KERNEL_SIZE, _arch_get(ref, 'STRIDE', 2))
        self.decoder = _Decoder(ARCH.DIM_LATENT, dec.CONV_CHAN, dec.CONV_LAYERS, dec.KERNEL_SIZE)
        self.posterior = _Posterior(self.dim_latent)

        self._cfg = _lib.Config(
            dim_latent=ARCH.DIM_LATENT, iters=ARCH.ITERS, slots=ARCH.SLOTS, img_size=ARCH.IMG_SIZE,
            img_channels=ARCH.IMG_CHANNELS, sigma=float(ARCH.SIGMA), layernorm=int(bool(ARCH.LAYERNORM)),
            stop_gradient=int(bool(self.use_stop_gradient)), encoding=_lib.encoding_bits(self.encodings),
            ref_conv_chan=ref.CONV_CHAN, ref_conv_layers=ref.CONV_LAYERS, ref_mlp_units=ref.MLP_UNITS,
            ref_kernel_size=ref.KERNEL_SIZE, ref_stride=_arch_get(ref, 'STRIDE', 2),
            dec_conv_chan=dec.CONV_CHAN, dec_conv_layers=dec.CONV_LAYERS, dec_kernel_size=dec.KERNEL_SIZE)

        # per-call state the reference keeps on self (iodine.py:36-52)
        self.lstm_hidden = None
        self.z = None
        self.mean = None
        self.mask_logits = None
        self.mask = None
        self.elbo_terms = None          # (n_elbo_calls, 3) = {ELBO, KL, LL} of the last call

        self._handle = None
        self._handle_device = None
        self._param_versions = None
        self._workspace = None
        self._ws_key = None
        self._options: Dict[str, float] = {}
        self._seed = 0                  # Philox key of the library's normal generator (manual_seed)
        self._draws = 0                 # Philox stream id: one per eps draw
        self._call_serial = 0           # bumped by every compute call; a backward must match the forward that saved state
        self._graph_stream = None
        self._graph_bufs: Dict[tuple, torch.Tensor] = {}

    # ---- bookkeeping identical to the reference -------------------------------------------------
    def get_input_size(self):
        """iodine.py:345-374."""
        size, latent = 0, 0
        e, c = self.encodings, self.img_channels
        if 'grad_post' in e: latent += 2 * self.dim_latent
        if 'posterior' in e: latent += 2 * self.dim_latent
        for name, n in (('image', c), ('means', c), ('mask', 1), ('mask_logits', 1), ('mask_posterior', 1),
                        ('grad_means', c), ('grad_mask', 1), ('likelihood', 1), ('leave_one_out_likelihood', 1),
                        ('coordinate', 2)):
            if name in e:
                size += n
        return size, latent

    # ---- library plumbing ---------------------------------------------------------------------
    def __del__(self):
        try:
            if self._handle is not None:
                _lib.lib().iodine_destroy(self._handle)
        except Exception:
            pass

    def _ordered_params(self):
        return [p for _, p in self.named_parameters()]

    def _ensure_handle(self, device: torch.device):
        if device.type != 'cuda':
            raise RuntimeError('iodine_amd.IODINE runs only on a ROCm device (gfx950); got tensors on '
                               f'{device}. There is no CPU fallback - move the module and inputs with .to("cuda").')
        L = _lib.lib()
        if self._handle is not None and self._handle_device != device:
            L.iodine_destroy(self._handle)
            self._handle, self._param_versions, self._workspace, self._ws_key = None, None, None, None
        if self._handle is None:
            with torch.cuda.device(device):
                h = C.c_void_p()
                _lib.check(L.iodine_create(C.byref(self._cfg), C.byref(h)), None, 'iodine_create')
            self._handle, self._handle_device = h, device
            for k, v in self._options.items():
                if k in _WRAPPER_OPTIONS:
                    continue
                _lib.check(L.iodine_set_option(h, k.encode(), v), h)
            names = []
            for i in range(L.iodine_num_params(h)):
                nm, nd, dims = C.c_char_p(), C.c_int(), (C.c_longlong * 4)()
                _lib.check(L.iodine_param_info(h, i, C.byref(nm), C.byref(nd), dims), h)
                names.append((nm.value.decode(), tuple(dims[:nd.value])))
            mine = [(n, tuple(p.shape)) for n, p in self.named_parameters()]
            if names != mine:
                raise RuntimeError(f'parameter table mismatch between module and library:\n{names}\n{mine}')
        return self._handle

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _launch(self, device, fn):
        """Run ``fn()`` (library calls on the current stream) with ``device`` current.  With option ``graph`` the calls go to
        a private non-default stream (the legacy null stream cannot be captured), ordered after / before the caller's."""
        with torch.cuda.device(device):
            if not self._options.get('graph'):
                return fn()
            if self._graph_stream is None or self._graph_stream.device != device:
                self._graph_stream = torch.cuda.Stream(device=device)
            cur = torch.cuda.current_stream()
            self._graph_stream.wait_stream(cur)
            with torch.cuda.stream(self._graph_stream):
                out = fn()
            cur.wait_stream(self._graph_stream)
            return out

    # With option ``graph`` the library keys its hipGraphs on the full argument tuple, device addresses included.  Tensors the
    # caching allocator hands out per call (inputs made contiguous, noise, outputs, the flat gradient buffer) would change that
    # key from step to step - every call an eager run or a re-capture instead of a replay.  So in graph mode the library only
    # ever sees persistent staging buffers owned by the module: inputs are copied in, outputs are cloned out.
    def _graph_on(self):
        return bool(self._options.get('graph'))

    def _gbuf(self, name, shape, device):
        key = (name, tuple(shape), str(device))
        t = self._graph_bufs.get(key)
        if t is None:
            t = self._graph_bufs[key] = torch.empty(tuple(shape), device=device, dtype=torch.float32)
        return t

    def _stage(self, name, t):
        """``t`` as the library should see it: itself, or (graph mode) its copy in the persistent buffer ``name``."""
        if t is None or not self._graph_on():
            return t
        buf = self._gbuf(name, t.shape, t.device)
        if buf.data_ptr() != t.data_ptr():
            buf.copy_(t)
        return buf

    def _out(self, name, shape, device):
        if self._graph_on():
            return self._gbuf(name, shape, device)
        return torch.empty(tuple(shape), device=device, dtype=torch.float32)

    def _own(self, t):
        """What the caller gets: the tensor itself, or (graph mode) a copy that the next call will not overwrite."""
        return t.clone() if (t is not None and self._graph_on()) else t

    def manual_seed(self, seed: int):
        """Seed of the library's generator for the ``eps=None`` draws (Philox4x32-10; give every rank its own seed)."""
        self._seed, self._draws = int(seed) & (2 ** 64 - 1), 0
        return self

    def mark_params_dirty(self):
        """Force a re-pack of the weights on the next call.  Needed only after writes that bypass autograd's version counter
        (``p.data.copy_(...)``, ``dist.broadcast(p.data, 0)``); optimizers, ``load_state_dict`` and ``.to()`` are seen."""
        self._param_versions = None

    def _sync_params(self, device):
        h = self._ensure_handle(device)
        params = self._ordered_params()
        for p in params:
            if p.device != device or p.dtype != torch.float32:
                raise RuntimeError('all parameters must be float32 on the same ROCm device as the input')
        versions = tuple((p.data_ptr(), p._version) for p in params)
        if versions != self._param_versions:
            keep = [p.detach().contiguous() for p in params]
            arr = (C.c_void_p * len(keep))(*[t.data_ptr() for t in keep])
            _lib.check(_lib.lib().iodine_set_params(h, self._stream(), arr, len(keep)), h, 'iodine_set_params')
            self._param_versions = versions
        return h

    def _ensure_workspace(self, h, B, mode, device):
        key = (B, mode, device)
        if self._ws_key == key:
            return
        need = _lib.lib().iodine_workspace_bytes(h, B, mode)
        if self._workspace is None or self._workspace.numel() < need or self._workspace.device != device:
            self._workspace = None
            self._workspace = torch.empty(need, dtype=torch.uint8, device=device)
        _lib.check(_lib.lib().iodine_set_workspace(h, C.c_void_p(self._workspace.data_ptr()),
                                                   self._workspace.numel()), h, 'iodine_set_workspace')
        self._ws_key = key

    def _check_x(self, x):
        if x.dim() != 4 or x.shape[1] != self.img_channels or x.shape[2] != self.img_size or x.shape[3] != self.img_size:
            raise RuntimeError(f'expected images of shape (B, {self.img_channels}, {self.img_size}, {self.img_size}), '
                               f'got {tuple(x.shape)}')
        return x.detach().to(torch.float32).contiguous()

    def _eps(self, eps, B, device):
        shape = (self.n_iters + 1, B, self.K, self.dim_latent)
        return self._normals(eps, shape, device)

    def _normals(self, eps, shape, device):
        if eps is None:
            out = self._out('eps', shape, device)
            self._launch(device, lambda: _lib.check(_lib.lib().iodine_randn(self._stream(), _lib.ptr(out), out.numel(),
                                                                          self._seed, self._draws), None, 'iodine_randn'))
            self._draws += 1
            return out
        if tuple(eps.shape) != shape:
            raise RuntimeError(f'eps must have shape {shape}, got {tuple(eps.shape)}')
        return self._stage('eps', eps.detach().to(device=device, dtype=torch.float32).contiguous())

    def debug_buffer(self, name: str, iteration: int = 0) -> torch.Tensor:
        """Copy of an internal workspace buffer of the last call (tests only)."""
        h = self._handle
        n = C.c_size_t()
        L = _lib.lib()
        _lib.check(L.iodine_debug_copy(h, self._stream(), name.encode(), iteration, None, 0, C.byref(n)), h)
        out = torch.empty(n.value, dtype=torch.float32, device=self._handle_device)
        _lib.check(L.iodine_debug_copy(h, self._stream(), name.encode(), iteration, C.c_void_p(out.data_ptr()),
                                       n.value, C.byref(n)), h)
        return out

    def set_option(self, key: str, value: float):
        """Debug/test options of the library (e.g. ``stop_after_iters``); applied to the live handle."""
        self._options[key] = float(value)
        if key in _WRAPPER_OPTIONS:                 # handled by this wrapper, unknown to the library
            return
        if self._handle is not None:
            _lib.check(_lib.lib().iodine_set_option(self._handle, key.encode(), float(value)), self._handle)
        if key in ('conv_precision', 'conv_variant'):
            self._param_versions = None          # the library keeps only the selected path's weight packs: re-send the parameters
        if key == 'wgrad_accum':
            self._ws_key = None                  # the workspace plan depends on it: ask the library again

    def profile_read(self, category: str, reset: bool = True):
        """(total_ms, launches) of one kernel category measured with HIP events (set_option('profile', 2); level 1 brackets the
        dominant ``conv_tile_*`` launches only)."""
        tot, cnt = C.c_double(), C.c_longlong()
        _lib.check(_lib.lib().iodine_profile_read(self._handle, category.encode(), C.byref(tot), C.byref(cnt),
                                                  int(reset)), self._handle)
        return tot.value, cnt.value

    # ---- inference: iodine.py:59-112 ------------------------------------------------------------
    def max_batch(self, training: bool = False) -> int:
        """Images one library call takes: the kernels index an activation tensor [B*K][pixels][channels] with 32-bit element
        offsets (iodine_api.cpp check_ready / iodine_train_forward).  Larger batches are run in chunks of at most this many
        images by the methods below - images are independent.  The ``batch_cap`` option lowers it (tests)."""
        P, K, T = self.img_size * self.img_size, self.K, self.n_iters
        cd, cr = int(self._cfg.dec_conv_chan), int(self._cfg.ref_conv_chan)
        lim = ((1 << 31) - 1) // (K * P * max(cd, cr, 20))
        if training:
            lim = min(lim, ((1 << 31) - 1) // (K * T * P * 20), ((1 << 31) - 1) // (K * T * max(((self.img_size - 1) // max(int(self._cfg.ref_stride), 1) + 1) ** 2, 1) * cr))
        cap = int(self._options.get('batch_cap', 0))
        return max(1, min(lim, cap) if cap > 0 else lim)

    @staticmethod
    def _chunks(B, cap):
        """Balanced split of B images into ceil(B / cap) runs: [(start, stop), ...]."""
        n = -(-B // cap)
        base, extra = divmod(B, n)
        out, s = [], 0
        for i in range(n):
            e = s + base + (1 if i < extra else 0)
            out.append((s, e))
            s = e
        return out

    def _merge_chunk_state(self, parts, sizes, x):
        """Per-call state and logger entries of a chunked call, as one call over the whole batch would have left them: tensors
        over the batch concatenated, batch means (ELBO terms) weighted by the chunks' sizes, image-0 entries from chunk 0."""
        B = float(sum(sizes))
        cat = lambda key: torch.cat([p[key] for p in parts], 0)
        self.z, self.mean, self.mask, self.mask_logits = cat('z'), cat('mean'), cat('mask'), cat('mask_logits')
        self.elbo_terms = sum(p['elbo_terms'] * (n / B) for p, n in zip(parts, sizes))
        logger.update(**parts[0]['logger0'])
        if self.elbo_terms.shape[0] > 0:
            logger.update(kl=self.elbo_terms[-1, 1], likelihood=self.elbo_terms[-1, 2])

    def _chunk_state(self):
        keep = ('image', 'pred') + tuple(f'mask_{i}' for i in range(self.K)) + tuple(f'pred_{i}' for i in range(self.K))
        return dict(z=self.z, mean=self.mean, mask=self.mask, mask_logits=self.mask_logits, elbo_terms=self.elbo_terms,
                    logger0={k: logger[k] for k in keep if k in logger})

    def _fetch_last_elbo(self, h, x, terms, count=None):
        """State the reference leaves on ``self`` after an ``elbo()`` call (iodine.py:171-187) and its logger entries
        (iodine.py:225-239): z, mean, mask, mask_logits of the whole batch and pred/image of image 0."""
        dev, B = x.device, x.shape[0]
        K, L, S = self.K, self.dim_latent, self.img_size
        f = dict(device=dev, dtype=torch.float32)
        z, mean = torch.empty((B, K, L), **f), torch.empty((B, K, 3, S, S), **f)
        mask, logits, pred = torch.empty((B, K, 1, S, S), **f), torch.empty((B, K, 1, S, S), **f), torch.empty((B, 3, S, S), **f)
        self._launch(dev, lambda: _lib.check(_lib.lib().iodine_last_elbo_outputs(
            h, self._stream(), B, _lib.ptr(z), _lib.ptr(mean), _lib.ptr(mask), _lib.ptr(logits), _lib.ptr(pred)),
            h, 'iodine_last_elbo_outputs'))
        self.z, self.mean, self.mask, self.mask_logits = z, mean, mask, logits
        logger.update(image=x[0], pred=pred[0], kl=terms[1], likelihood=terms[2])
        logger.update(**{f'mask_{i}': mask[0, i, 0] for i in range(K)})
        logger.update(**{f'pred_{i}': mean[0, i] for i in range(K)})

    def _fetch_posterior(self, h, B, dev):
        """``self.posterior.mean / logvar`` = lambda_T, what ``Gaussian.update`` (iodine.py:636-645) leaves on the reference's
        module after ``forward``: a following ``model.elbo(x)`` samples from it."""
        pm = torch.empty((B, self.K, self.dim_latent), device=dev, dtype=torch.float32)
        plv = torch.empty_like(pm)
        self._launch(dev, lambda: _lib.check(_lib.lib().iodine_last_posterior(h, self._stream(), B, _lib.ptr(pm), _lib.ptr(plv)),
                                             h, 'iodine_last_posterior'))
        self.posterior.mean, self.posterior.logvar = pm, plv

    @torch.no_grad()
    def _reconstruct(self, x, eps, want_images=True):
        x = self._check_x(x)
        dev, B = x.device, x.shape[0]
        cap = self.max_batch()
        if B > cap:                                   # chunks of independent images; eps (T+1, B, K, L) is cut along B
            outs, parts, sizes, pms, plvs = [], [], [], [], []
            for s, e in self._chunks(B, cap):
                outs.append(self._reconstruct(x[s:e], None if eps is None else eps[:, s:e], want_images))
                parts.append(self._chunk_state()); sizes.append(e - s)
                pms.append(self.posterior.mean); plvs.append(self.posterior.logvar)
            self.posterior.mean, self.posterior.logvar = torch.cat(pms, 0), torch.cat(plvs, 0)
            self._merge_chunk_state(parts, sizes, x)
            return tuple(None if outs[0][j] is None else torch.cat([o[j] for o in outs], 0) for j in range(4))
        h = self._sync_params(dev)
        self._ensure_workspace(h, B, 0, dev)
        eps = self._eps(eps, B, dev)
        xs = self._stage('x', x)
        K, L, S, T = self.K, self.dim_latent, self.img_size, self.n_iters
        pred = self._out('r.pred', (B, 3, S, S), dev) if want_images else None
        mask = self._out('r.mask', (B, K, 1, S, S), dev) if want_images else None
        mean = self._out('r.mean', (B, K, 3, S, S), dev) if want_images else None
        z = self._out('r.z', (B, K, L), dev)
        pm, plv = self._out('r.pm', (B, K, L), dev), self._out('r.plv', (B, K, L), dev)
        stop = int(self._options.get('stop_after_iters', -1))
        n_it = stop if 0 <= stop <= T else T
        elbo = self._out('r.elbo', (n_it, 3), dev)
        self._call_serial += 1
        self._launch(dev, lambda: _lib.check(_lib.lib().iodine_reconstruct(
            h, self._stream(), B, _lib.ptr(xs), _lib.ptr(eps), _lib.ptr(pred), _lib.ptr(mask), _lib.ptr(mean), _lib.ptr(z),
            _lib.ptr(pm), _lib.ptr(plv), _lib.ptr(elbo)), h, 'iodine_reconstruct'))
        pred, mask, mean, z, pm, plv, elbo = (self._own(t) for t in (pred, mask, mean, z, pm, plv, elbo))
        self.posterior.mean, self.posterior.logvar, self.elbo_terms = pm, plv, elbo
        if n_it > 0:
            self._fetch_last_elbo(h, x, elbo[-1])        # what the reference's last elbo() call left behind
        return pred, mask, mean, z

    def encode(self, x, eps=None):
        """z (B, K, L) after T refinement iterations.  iodine.py:73-105."""
        return self._reconstruct(x, eps, want_images=False)[3]

    def reconstruct(self, x, eps=None):
        """pred (B,3,S,S), mask (B,K,1,S,S), mean (B,K,3,S,S).  iodine.py:107-112."""
        pred, mask, mean, _ = self._reconstruct(x, eps)
        return pred, mask, mean

    @torch.no_grad()
    def decode(self, z):
        """iodine.py:59-71."""
        z = z.detach().to(torch.float32).contiguous()
        dev, B = z.device, z.shape[0]
        cap = self.max_batch()
        if B > cap:
            outs = [self.decode(z[s:e]) for s, e in self._chunks(B, cap)]
            return tuple(torch.cat([o[j] for o in outs], 0) for j in range(3))
        h = self._sync_params(dev)
        self._ensure_workspace(h, B, 0, dev)
        K, S = self.K, self.img_size
        z = self._stage('d.z', z)
        pred, mask, mean = self._out('r.pred', (B, 3, S, S), dev), self._out('r.mask', (B, K, 1, S, S), dev), self._out('r.mean', (B, K, 3, S, S), dev)
        self._call_serial += 1
        self._launch(dev, lambda: _lib.check(_lib.lib().iodine_decode(h, self._stream(), B, _lib.ptr(z), _lib.ptr(pred),
                                                                      _lib.ptr(mask), _lib.ptr(mean)), h, 'iodine_decode'))
        return self._own(pred), self._own(mask), self._own(mean)

    @torch.no_grad()
    def elbo(self, x, eps=None):
        """Single-pass ELBO (iodine.py:161-241): one sample from the current posterior (``self.posterior.mean / logvar`` as
        left by the last call for this batch size; otherwise the initial posterior of ``init_unit``, iodine.py:607-618),
        decode, mixture log-likelihood minus KL.  Sets ``self.z / mean / mask / mask_logits`` and the logger entries like the
        reference.  ``eps`` (B, K, L) replaces the ``torch.randn_like`` draw.  Returns the scalar ELBO (no autograd graph:
        the gradients the reference takes from it are what reconstruct / forward compute in closed form)."""
        x = self._check_x(x)
        dev, B = x.device, x.shape[0]
        cap = self.max_batch()
        if B > cap:
            pm0, plv0 = self.posterior.mean, self.posterior.logvar
            whole = pm0 is not None and plv0 is not None and tuple(pm0.shape) == (B, self.K, self.dim_latent) and pm0.device == dev
            parts, sizes = [], []
            for s, e in self._chunks(B, cap):
                # the chunk's slice of the current posterior (or the initial posterior, as for a whole batch)
                self.posterior.mean, self.posterior.logvar = (pm0[s:e], plv0[s:e]) if whole else (None, None)
                self.elbo(x[s:e], None if eps is None else eps[s:e])
                parts.append(self._chunk_state()); sizes.append(e - s)
            self.posterior.mean, self.posterior.logvar = pm0, plv0
            self._merge_chunk_state(parts, sizes, x)
            return self.elbo_terms[0, 0]
        h = self._sync_params(dev)
        self._ensure_workspace(h, B, 0, dev)
        shape = (B, self.K, self.dim_latent)
        eps = self._normals(eps, shape, dev)
        pm, plv = self.posterior.mean, self.posterior.logvar
        if pm is None or plv is None or tuple(pm.shape) != shape or pm.device != dev:
            pm = plv = None
        else:
            pm = self._stage('e.pm', pm.detach().to(torch.float32).contiguous())
            plv = self._stage('e.plv', plv.detach().to(torch.float32).contiguous())
        terms = self._out('e.terms', (3,), dev)
        xs = self._stage('x', x)
        self._call_serial += 1
        self._launch(dev, lambda: _lib.check(_lib.lib().iodine_elbo(h, self._stream(), B, _lib.ptr(xs), _lib.ptr(pm), _lib.ptr(plv),
                                                                    _lib.ptr(eps), _lib.ptr(terms)), h, 'iodine_elbo'))
        terms = self._own(terms)
        self.elbo_terms = terms.view(1, 3)
        self._fetch_last_elbo(h, x, terms)
        return terms[0]

    # ---- training: iodine.py:115-158 + lib/engine/train.py:60-63 -------------------------------------
    def forward(self, x, eps=None):
        """-sum_i (i+1)/(T+1) ELBO_i, differentiable wrt every parameter."""
        x = self._check_x(x)
        if x.shape[0] > self.max_batch(training=True):
            loss, elbo_iter = _ChunkedTrainStep.apply(self, x, eps, *self._ordered_params())
            self.elbo_terms = elbo_iter
            return loss
        eps = self._eps(eps, x.shape[0], x.device)
        loss, elbo_iter = _TrainStep.apply(self, x, eps, *self._ordered_params())
        self.elbo_terms = elbo_iter
        with torch.no_grad():
            h, dev = self._handle, x.device
            self._fetch_last_elbo(h, x, elbo_iter[-1])                                 # final elbo(): iodine.py:226-239
            self._fetch_posterior(h, x.shape[0], dev)
            stats = torch.empty((2,), device=dev, dtype=torch.float32)
            self._launch(dev, lambda: _lib.check(_lib.lib().iodine_logger_scalars(h, self._stream(), _lib.ptr(stats)), h))
            logger.update(init_mean=stats[0], init_logvar=stats[1])                    # iodine.py:156-157
        return loss

    def _train_forward(self, x, eps):
        dev, B = x.device, x.shape[0]
        h = self._sync_params(dev)
        self._ensure_workspace(h, B, 1, dev)
        loss = self._out('t.loss', (), dev)
        elbo_iter = self._out('t.elbo', (self.n_iters + 1, 3), dev)
        xs, eps = self._stage('x', x), self._stage('eps', eps)
        self._call_serial += 1
        self._launch(dev, lambda: _lib.check(_lib.lib().iodine_train_forward(h, self._stream(), B, _lib.ptr(xs), _lib.ptr(eps),
                                                                             _lib.ptr(loss), _lib.ptr(elbo_iter)),
                                             h, 'iodine_train_forward'))
        return self._own(loss), self._own(elbo_iter)

    def _train_chunked(self, x, eps):
        """Forward + backward of every chunk (see _ChunkedTrainStep): returns the batch loss, the (T+1, 3) ELBO terms of the whole
        batch and d loss / d parameters as one flat buffer in named_parameters() order."""
        dev, B = x.device, x.shape[0]
        flat = self._out('t.flat', (sum(p.numel() for p in self._ordered_params()),), dev)
        loss, terms, parts, sizes, pms, plvs = None, None, [], [], [], []
        for c, (s, e) in enumerate(self._chunks(B, self.max_batch(training=True))):
            xc = x[s:e].contiguous()
            ec = self._eps(None if eps is None else eps[:, s:e], e - s, dev)
            lc, tc = self._train_forward(xc, ec)
            h = self._handle
            w = torch.full((), (e - s) / float(B), device=dev, dtype=torch.float32)
            ws = self._stage('t.gl', w)
            self._launch(dev, lambda: _lib.check(_lib.lib().iodine_train_backward_flat(h, self._stream(), _lib.ptr(ws), _lib.ptr(flat),
                                                                                       1 if c else 0), h, 'iodine_train_backward'))
            with torch.no_grad():
                self._fetch_last_elbo(h, xc, tc[-1])
                self._fetch_posterior(h, e - s, dev)
                pms.append(self.posterior.mean); plvs.append(self.posterior.logvar)
                self.elbo_terms = tc
                parts.append(self._chunk_state()); sizes.append(e - s)
                if c == 0:
                    stats = torch.empty((2,), device=dev, dtype=torch.float32)
                    self._launch(dev, lambda: _lib.check(_lib.lib().iodine_logger_scalars(h, self._stream(), _lib.ptr(stats)), h))
                    logger.update(init_mean=stats[0], init_logvar=stats[1])
            self._call_serial += 1                  # this chunk's saved forward is consumed
            loss = lc * w if loss is None else loss + lc * w
        with torch.no_grad():
            self._merge_chunk_state(parts, sizes, x)
            self.posterior.mean, self.posterior.logvar = torch.cat(pms, 0), torch.cat(plvs, 0)
        return loss, self.elbo_terms, self._own(flat)

    def _train_backward(self, grad_loss, serial):
        if serial != self._call_serial:
            raise RuntimeError('IODINE: backward of a stale forward - the library keeps the saved state of ONE forward pass and '
                               'another forward / reconstruct / decode / elbo call has re-used it since (the reference would '
                               'hold a second autograd graph; call loss.backward() before the next model call)')
        h, dev = self._handle, self._handle_device
        params = self._ordered_params()
        sizes = [p.numel() for p in params]
        flat = self._out('t.flat', (sum(sizes),), dev)
        gl = self._stage('t.gl', grad_loss.detach().to(device=dev, dtype=torch.float32).contiguous())
        self._launch(dev, lambda: _lib.check(_lib.lib().iodine_train_backward_flat(h, self._stream(), _lib.ptr(gl), _lib.ptr(flat), 0),
                                             h, 'iodine_train_backward'))
        self._call_serial += 1                      # the saved forward is consumed (no retain_graph)
        flat = self._own(flat)                      # graph mode: autograd may keep what we return as .grad; never the staging buffer
        views, off = [], 0
        for p, n in zip(params, sizes):
            views.append(flat[off:off + n].view_as(p))
            off += n
        return views


def arch_namespace(dim_latent, iters, slots, img_size, ref, dec, sigma=0.10, layernorm=True,
                   encoding=_lib.ENC_ORDER, kernels=(3, 3), ref_stride=2):
    """Build an ``ARCH``-shaped namespace (the yacs node of lib/config/defaults.py:35-100) from plain
    values; ref = (CONV_CHAN, CONV_LAYERS, MLP_UNITS), dec = (CONV_CHAN, CONV_LAYERS), kernels = (REF, DEC) KERNEL_SIZE."""
    from types import SimpleNamespace as NS
    return NS(DIM_LATENT=dim_latent, ITERS=iters, SLOTS=slots, ENCODING=list(encoding), IMG_CHANNELS=3,
              IMG_SIZE=img_size, SIGMA=sigma, LAYERNORM=layernorm, STOP_GRADIENT=False,
              REF=NS(CONV_CHAN=ref[0], CONV_LAYERS=ref[1], MLP_UNITS=ref[2], KERNEL_SIZE=kernels[0], STRIDE=ref_stride),
              DEC=NS(CONV_CHAN=dec[0], CONV_LAYERS=dec[1], KERNEL_SIZE=kernels[1]))


def clevr6_arch(slots=7, iters=5):
    """configs/clevr6_prop.yaml:26-45."""
    return arch_namespace(64, iters, slots, 128, (64, 4, 256), (64, 4))


def dsprites_arch(slots=6, iters=5):
    """configs/dsprites_noclip.yaml:26-45."""
    return arch_namespace(16, iters, slots, 64, (32, 3, 128), (32, 5))
