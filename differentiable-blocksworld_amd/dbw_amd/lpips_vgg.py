"""The perceptual criterion of the shipped configs: LPIPS with the VGG16 backbone, as the reference builds it in
src/model/loss.py:32-40 (`lpips.LPIPS(net='vgg')`, called with normalize=True, mean over the batch), written in plain torch so that it
runs on MIOpen convolutions next to the HIP render path -- outside of it, as SURVEY.md 8(a) A10 prescribes.

Neither the `lpips` package nor any weights exist in this environment, so REAL-WEIGHT PARITY IS UNPINNED.  What is pinned is the
architecture: oracle/lpips_ref.py restates the published forward of lpips 0.1.4 independently, tests/golden/lpips_random.npz freezes
it on seeded random weights, and this module reproduces that fixture -- forward and gradient, CPU and GPU -- when loaded with the same
weights (tests/test_lpips.py); the model's perceptual term is checked against the restatement too.  Sources of the constants:
  * the scaling layer constants and the layer taps (relu1_2, relu2_2, relu3_3, relu4_3, relu5_3 of VGG16) are those of the published
    lpips 0.1.4 package (environment.yml:29), restated from its documentation;
  * `load_weights` takes the two state dicts a user has to bring: torchvision's `vgg16().features` ('0.weight', '0.bias', '2.weight', ...)
    and lpips' linear heads ('lin0.model.1.weight' ... 'lin4.model.1.weight', each (1, C, 1, 1), non-negative);
  * without weights the forward refuses to run unless `allow_random_init=True` (shape / throughput checks only).

Usage: `model.set_perceptual(LPIPSVGG().load_weights(vgg_sd, lin_sd).to(device))`.

One option changes the cost, not the value (measured on MI355X at 4 + 4 images of 400x300: 12.5 ms forward + backward as the reference
runs it -- 878 GFLOP of fp32 convolutions; 9.0 ms with it): `cache_targets(imgs_all)`.  The training images never change and the network is
frozen, so their normalised features are constants -- computed once for every view (58 MB per 400x300 view) and gathered by
`forward(imgs, rec, view_ids=...)`: a third of the FLOPs gone (the target branch's forward; it never had a backward).  The Trainer does this
when the features fit (trainer.py).  (channels_last weights and activations were measured too: 11.8 ms where MIOpen had searched its
kernels in the same process, 20.5 ms where it had not -- not offered.)"""
import torch
from torch import nn
import torch.nn.functional as F

# VGG16 `features`: index of every conv layer and its (in, out) channels; taps after the ReLU that follows the listed conv
_VGG16_CONVS = [(0, 3, 64), (2, 64, 64), (5, 64, 128), (7, 128, 128), (10, 128, 256), (12, 256, 256), (14, 256, 256),
                (17, 256, 512), (19, 512, 512), (21, 512, 512), (24, 512, 512), (26, 512, 512), (28, 512, 512)]
_POOL_BEFORE = {5, 10, 17, 24}                 # a 2x2 max-pool sits in front of these convs (features.4, 9, 16, 23)
_TAPS = {2: 0, 7: 1, 14: 2, 21: 3, 28: 4}      # relu1_2, relu2_2, relu3_3, relu4_3, relu5_3
_CHANNELS = [64, 128, 256, 512, 512]


def _f32_nchw(t):
    """Whether the HIP passes below may be handed this tensor's data_ptr(): they index float32 in (N, C, h, w) order -- a half-precision
    output of a convolution under torch.autocast, or a channels_last one, must go through torch's own ops instead."""
    return t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()


class _FusedHead(torch.autograd.Function):
    """One tap's head on the device in one pass each way (csrc/lpips_head.hip, include/dbw_hip.h: dbw_lpips_head_*): per-image values (N,)
    from the reconstruction's tap f (N,C,h,w), the targets' unit-normalised tap (rows `ids` of it when given) and the 1x1 head's weights."""

    @staticmethod
    def forward(ctx, f, target_unit, ids, w):
        from . import _lib
        f = f.contiguous()
        N, C, h, wd = f.shape
        nb = _lib.load().dbw_lpips_head_blocks(N, h * wd)
        partial = torch.empty(N, nb, dtype=torch.float32, device=f.device)
        with torch.cuda.device(f.device):
            _lib.call('dbw_lpips_head_fwd', f.data_ptr(), target_unit.data_ptr(), ids.data_ptr() if ids is not None else 0, w.data_ptr(), N, target_unit.shape[0], C, h * wd,
                      partial.data_ptr(), torch.cuda.current_stream(f.device).cuda_stream)
        ctx.save_for_backward(f, target_unit, w, *([ids] if ids is not None else []))
        return partial.sum(1)

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        f, target_unit, w = ctx.saved_tensors[:3]
        ids = ctx.saved_tensors[3] if len(ctx.saved_tensors) > 3 else None
        N, C, h, wd = f.shape
        g = g.contiguous().float()
        gf = torch.empty_like(f)
        with torch.cuda.device(f.device):
            _lib.call('dbw_lpips_head_bwd', f.data_ptr(), target_unit.data_ptr(), ids.data_ptr() if ids is not None else 0, w.data_ptr(), N, target_unit.shape[0], C, h * wd,
                      g.data_ptr(), gf.data_ptr(), torch.cuda.current_stream(f.device).cuda_stream)
        return gf, None, None, None


class _BiasReLU(torch.autograd.Function):
    """max(x + bias[c], 0) in place on the output of a convolution that was run without its bias: one pass (dbw_bias_relu) where torch
    runs the bias add and the clamp as two; the bias is frozen (loss.py:36-37): gradient to x only."""

    @staticmethod
    def forward(ctx, x, bias):
        from . import _lib
        N, C, h, w = x.shape
        with torch.cuda.device(x.device):
            _lib.call('dbw_bias_relu', x.data_ptr(), bias.data_ptr(), N, C, h * w, x.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream)
        ctx.mark_dirty(x)
        ctx.save_for_backward(x)
        return x

    @staticmethod
    def backward(ctx, g):
        y, = ctx.saved_tensors
        return torch.ops.aten.threshold_backward(g.contiguous(), y, 0), None


class _MaxPool2(torch.autograd.Function):
    """F.max_pool2d(x, 2, 2) without an index tensor: the backward finds each window's maximum again (dbw_maxpool2_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, x):
        from . import _lib
        x = x.contiguous()
        N, C, h, w = x.shape
        y = torch.empty(N, C, h // 2, w // 2, dtype=x.dtype, device=x.device)
        with torch.cuda.device(x.device):
            _lib.call('dbw_maxpool2_fwd', x.data_ptr(), N * C, h, w, y.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        x, = ctx.saved_tensors
        N, C, h, w = x.shape
        g = g.contiguous()
        gx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.call('dbw_maxpool2_bwd', x.data_ptr(), g.data_ptr(), N * C, h, w, gx.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream)
        return gx


class LPIPSVGG(nn.Module):
    def __init__(self, allow_random_init=False):
        super().__init__()
        self.convs = nn.ModuleDict({str(i): nn.Conv2d(cin, cout, 3, padding=1) for i, cin, cout in _VGG16_CONVS})
        self.lins = nn.ModuleList([nn.Conv2d(c, 1, 1, bias=False) for c in _CHANNELS])
        self.register_buffer('shift', torch.tensor([-.030, -.088, -.188]).view(1, 3, 1, 1))
        self.register_buffer('scale', torch.tensor([.458, .448, .450]).view(1, 3, 1, 1))
        self.loaded = False
        self.allow_random_init = allow_random_init
        self.fused_head = True            # CUDA tensors: the head (normalise, difference, 1x1 head, spatial mean) as one HIP pass per tap and
                                          # direction, bias + ReLU and the 2x2 max-pools as one pass each (csrc/lpips_head.hip)
        self.target_cache = None          # per tap: (V, C, h, w) normalised features of the V training images (cache_targets)
        if allow_random_init:
            with torch.no_grad():
                for lin in self.lins:
                    lin.weight.abs_()                                # the published heads are non-negative
        for p in self.parameters():
            p.requires_grad = False                                  # loss.py:36-37

    def load_weights(self, vgg_features_state, lin_state):
        """vgg_features_state: state dict of torchvision `vgg16().features`; lin_state: lpips' `lin{k}.model.1.weight` tensors."""
        with torch.no_grad():
            for i, _, _ in _VGG16_CONVS:
                self.convs[str(i)].weight.copy_(vgg_features_state[f'{i}.weight'])
                self.convs[str(i)].bias.copy_(vgg_features_state[f'{i}.bias'])
            for k, lin in enumerate(self.lins):
                lin.weight.copy_(lin_state[f'lin{k}.model.1.weight'])
        self.loaded = True
        self.target_cache = None          # (features of the previous weights)
        self._cache_key = None
        return self

    @staticmethod
    def _unit(f):
        return f / (f.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)

    def target_bytes(self, H, W, n_views=1):
        """Bytes cache_targets needs for n_views images of H x W."""
        total, h, w = 0, H, W
        for k, c in enumerate(_CHANNELS):
            if k:
                h, w = h // 2, w // 2
            total += c * h * w * 4
        return total * n_views

    @torch.no_grad()
    def cache_targets(self, imgs_all, chunk=8):
        """imgs_all (V,3,H,W) in [0, 1]: the training images, in the order the `view_ids` of forward() index them.  Valid as long as the
        images and the (frozen) weights stay what they are; `cache_targets(None)` drops the cache."""
        if imgs_all is None:
            self.target_cache, self._cache_key = None, None
            return self
        if not (self.loaded or self.allow_random_init):
            raise RuntimeError('LPIPSVGG has no weights: bring torchvision vgg16 features + lpips linear heads (load_weights)')
        # (each tap's (V, C, h, w) buffer is allocated once and filled chunk by chunk: a list of parts + torch.cat holds the cache twice at its peak)
        V, cache = imgs_all.shape[0], None
        for a in range(0, V, chunk):
            feats = self.features(imgs_all[a:a + chunk] * 2 - 1)
            if cache is None:
                cache = [torch.empty((V,) + tuple(f.shape[1:]), dtype=f.dtype, device=f.device) for f in feats]
            for k, f in enumerate(feats):
                cache[k][a:a + f.shape[0]] = self._unit(f)
        self.target_cache = cache
        # whose features these are: forward(view_ids=...) refuses a cache built on other images (a second Trainer on other views)
        self._cache_key = (imgs_all.data_ptr(), tuple(imgs_all.shape), imgs_all._version) if V else None
        return self

    def cache_matches(self, imgs_all):
        """Whether the cache was built from exactly this tensor (address, shape, version counter)."""
        return self.target_cache is not None and getattr(self, '_cache_key', None) == (imgs_all.data_ptr(), tuple(imgs_all.shape), imgs_all._version)

    def features(self, x):
        x = (x - self.shift) / self.scale
        taps = []
        fused = x.is_cuda and self.fused_head and x.dtype == torch.float32      # the layers between the convolutions as single HIP passes
        for i, _, _ in _VGG16_CONVS:
            conv = self.convs[str(i)]
            if i in _POOL_BEFORE:
                x = _MaxPool2.apply(x) if (fused and _f32_nchw(x)) else F.max_pool2d(x, 2, 2)
            # (decided on the tensor each kernel is actually handed: under torch.autocast the convolution returns half precision whatever
            # its input was, and an in-place float32 pass over that buffer would write twice its size)
            y = F.conv2d(x, conv.weight, None, padding=1) if fused else None
            if y is not None and _f32_nchw(y) and conv.bias.dtype == torch.float32:
                x = _BiasReLU.apply(y, conv.bias)
            elif y is not None:
                x = F.relu(y + conv.bias.to(y.dtype).view(1, -1, 1, 1))
            else:
                x = F.relu(conv(x))
            if i in _TAPS:
                taps.append(x)
        return taps

    def forward(self, imgs, rec, view_ids=None):
        """imgs, rec (B,3,H,W) in [0, 1] -> scalar: mean over the batch of sum_l mean_hw lin_l((n(f_l(a)) - n(f_l(b)))^2), n = unit
        normalisation along channels.  view_ids (B,) integer tensor on the device of `rec` + a cache (cache_targets): the targets' features
        are gathered from the cache instead of being recomputed from `imgs` (which is then not read)."""
        if not (self.loaded or self.allow_random_init):
            raise RuntimeError('LPIPSVGG has no weights: bring torchvision vgg16 features + lpips linear heads (load_weights)')
        cached = view_ids is not None and self.target_cache is not None
        if cached:
            if view_ids.shape[0] != rec.shape[0]:
                raise ValueError(f'{view_ids.shape[0]} view ids for {rec.shape[0]} images')
            view_ids = view_ids.to(device=rec.device, dtype=torch.int64).contiguous()          # (a loader may hand them over on the host)
        if not cached:
            na_all = [self._unit(fa) for fa in self.features(imgs * 2 - 1)]          # normalize=True
        fb_all = self.features(rec * 2 - 1)
        heads_ok = all(fb.is_cuda and fb.dtype == torch.float32 for fb in fb_all) and \
            all(t.dtype == torch.float32 for t in (self.target_cache if cached else na_all))
        if rec.is_cuda and self.fused_head and heads_ok and not any(t.requires_grad for t in (na_all if not cached else [])):
            # the head of every tap in one pass each way on the device (csrc/lpips_head.hip); the targets' rows are read in place
            ids = view_ids if cached else None
            per = sum(_FusedHead.apply(fb, (self.target_cache[k] if cached else na_all[k]).contiguous(), ids, lin.weight.detach().reshape(-1).contiguous())
                      for k, (fb, lin) in enumerate(zip(fb_all, self.lins)))
            return per.mean()
        if cached:
            na_all = [c.index_select(0, view_ids) for c in self.target_cache]
        total = 0
        for na, fb, lin in zip(na_all, fb_all, self.lins):
            total = total + lin((na - self._unit(fb)) ** 2).mean((2, 3), keepdim=True)
        return total.mean()
