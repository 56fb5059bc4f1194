"""torch.autograd Functions whose forward AND backward are the HIP kernels of the K3 chain.

Used by the mixed-16-bit training step (`TrainStep(precision='bf16', hip_backward=True)`): the coarse layers' linear attention
(`HipLinearAttention`: K2 forward, `gf_linear_attention_backward`) and the linears, LayerNorms and
activations of the encoder layers (loftr_module/transformer.py:45-60, geo_transformer/transformer.py:49-66) and of
FinePreprocess (fine_preprocess.py:61-72) run `gf_linear` / `gf_layernorm_forward` forward and
`gf_linear` (dX = dY W), `gf_linear_wgrad` (dW = dY^T X), `gf_layernorm_backward`, `gf_activation_backward` backward.
Master parameters stay fp32: a Function receives the fp32 parameter, casts it to the activations' 16-bit type for the
kernels (what torch.autocast does for nn.Linear) and returns an fp32 gradient.

What is saved for backward: the 16-bit inputs of every linear, the pre-LayerNorm rows with their (mean, rstd), the
activation OUTPUTS (ReLU / Tanh derivatives are functions of the output) - the same tensors autograd keeps.
"""
import torch

from .. import ops


class WeightCache:
    """16-bit copies of the fp32 master weights and their transposes, made once per step (a weight is used by several layer
    calls - both images, forward and backward - and every cast or transpose is a launch the host pays for).  An entry is
    valid for one VERSION of one storage: the hit check compares the parameter's identity, its `_version` counter (every
    in-place write - optimizer.step(), load_state_dict's copy_ - bumps it) and its data pointer (`p.data = ...`), so a
    training loop that uses the HIP Functions without `TrainStep` (functional.set_hip_backward is public) never runs on
    stale 16-bit weights.  `TrainStep` still clears the cache behind the optimizer step to drop the dead copies early."""

    def __init__(self):
        self._w, self._wt = {}, {}

    def clear(self):
        self._w.clear()
        self._wt.clear()

    def cast(self, w, dtype):
        key = (id(w), dtype)
        stamp = (w._version, w.data_ptr())
        hit = self._w.get(key)
        if hit is None or hit[0] is not w or hit[2] != stamp:
            if hit is not None:
                self._wt.pop(id(hit[1]), None)                         # the transposed copy of the superseded cast
            hit = (w, w.detach().to(dtype), stamp)
            self._w[key] = hit
        return hit[1]

    def transposed(self, w16):
        key = id(w16)
        hit = self._wt.get(key)
        if hit is None or hit[0] is not w16:
            hit = (w16, w16.t().contiguous())
            self._wt[key] = hit
        return hit[1]


WEIGHTS = WeightCache()


class HipLinear(torch.autograd.Function):
    """y = act([x | x2] @ w.T), no bias; x [..., k1], x2 [..., k2] or None (the two halves of torch.cat([x, message], 2)),
    w fp32 or 16-bit [n, k1 + k2]; act in (None, 'relu', 'tanh')."""

    @staticmethod
    def forward(ctx, x, w, x2=None, act=None):
        w16 = w if w.dtype == x.dtype else WEIGHTS.cast(w, x.dtype)
        epi = {None: ops.EPI_NONE, 'relu': ops.EPI_RELU, 'tanh': ops.EPI_TANH}[act]
        y = ops.linear(x, w16, a2=x2, epilogue=epi)
        ctx.act, ctx.k1, ctx.has2, ctx.wdtype = act, x.shape[-1], x2 is not None, w.dtype
        ctx.save_for_backward(x, x2 if x2 is not None else x.new_empty(0), w16, y if act is not None else x.new_empty(0))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, x2, w16, y = ctx.saved_tensors
        dz = ops.activation_backward(dy, y, ctx.act) if ctx.act is not None else dy.contiguous()
        need_x, need_w, need_x2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has2 and ctx.needs_input_grad[2]
        dx = dx2 = dw = None
        if need_x or need_x2:
            dfull = ops.linear(dz, WEIGHTS.transposed(w16))            # [..., k1 + k2] = dZ W
            dx = dfull[..., :ctx.k1] if need_x else None
            dx2 = dfull[..., ctx.k1:] if need_x2 else None
        if need_w:
            dw = torch.empty(w16.shape, dtype=torch.float32, device=w16.device)
            ops.linear_wgrad(dz, x, out=dw[:, :ctx.k1])
            if ctx.has2:
                ops.linear_wgrad(dz, x2, out=dw[:, ctx.k1:])
            if ctx.wdtype != torch.float32:
                dw = dw.to(ctx.wdtype)
        return dx, dw, dx2, None


class HipLayerNorm(torch.autograd.Function):
    """nn.LayerNorm over the last dimension (128, 256 or 512 channels), 16-bit rows, fp32 gamma / beta and statistics."""

    @staticmethod
    def forward(ctx, y, gamma, beta, eps=1e-5):
        g32, b32 = gamma.detach().float(), beta.detach().float()
        out, stats = ops.layernorm_forward(y, g32, b32, eps)
        ctx.save_for_backward(y, stats, g32)
        ctx.pdtype = gamma.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        y, stats, g32 = ctx.saved_tensors
        dy, dg, db = ops.layernorm_backward(dout, y, stats, g32)
        return dy, dg.to(ctx.pdtype), db.to(ctx.pdtype), None


class HipLinearAttention(torch.autograd.Function):
    """LinearAttention.forward (linear_attention.py:21-51) on [N, L, C] / [N, S, C] 16-bit tensors with heads of 32 channels:
    forward = K2 (gf_linear_attention), backward = gf_linear_attention_backward (csrc/k_train.hip)."""

    @staticmethod
    def forward(ctx, q, k, v, nhead, q_mask=None, kv_mask=None):
        out = ops.linear_attention(q, k, v, nhead, q_mask, kv_mask)
        ctx.nhead = nhead
        ctx.masks = (q_mask, kv_mask)
        ctx.save_for_backward(q, k, v)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v = ctx.saved_tensors
        dq, dk, dv = ops.linear_attention_backward(q, k, v, dout.contiguous(), ctx.nhead, *ctx.masks)
        return dq, dk, dv, None, None, None


class HipFullAttention(torch.autograd.Function):
    """FullAttention.forward (geo_attention.py:72-101, no masks) as GeoTransformer's 'self' branch calls it (transformer.py:111-124): q [N,L,256]
    against the projected inlier rows k, v [N,S,256], 4 heads of 64, 16-bit tensors.  forward = gf_full_attention_train_forward (keeps the
    rows' log-sum-exp), backward = gf_full_attention_backward (csrc/k4_attention_train.hip: flash form, bit-reproducible)."""

    @staticmethod
    def forward(ctx, q, k, v, nhead):
        out, lse = ops.full_attention_train_forward(q, k, v, nhead)
        ctx.nhead = nhead
        ctx.save_for_backward(q, k, v, out, lse)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        dq, dk, dv = ops.full_attention_backward(q, k, v, out, dout, lse, ctx.nhead)
        return dq, dk, dv, None


class HipWindowLinearAttention(torch.autograd.Function):
    """LinearAttention.forward on the fine level's windows ([Nw, Lw <= 32, 128], 8 heads of 16, no masks; full_model.py:97-98):
    forward = K2's window form (la_window_mfma), backward = gf_window_linear_attention_backward (csrc/k_train.hip)."""

    @staticmethod
    def forward(ctx, q, k, v, nhead):
        out = ops.linear_attention(q, k, v, nhead)
        ctx.save_for_backward(q, k, v)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v = ctx.saved_tensors
        dq, dk, dv = ops.window_linear_attention_backward(q, k, v, dout)
        return dq, dk, dv, None


_GATHER_K5_BACKWARD = [True]       # K5's backward as a gather along the inverse window table (False: the fp32 atomic scatter, for A/B)


class HipWindowCrossAttention(torch.autograd.Function):
    """FullAttention.forward over the 25 keys of each query's window (geo_attention.py:72-101 as GeoTransformer's 'cross' branch
    calls it, transformer.py:125-139) on the PROJECTED maps: q [N,L,256], kmap / vmap [N,S,256] = k_proj / v_proj of every token of
    the other image, win int32 [N,L,25] (cell of the other image per window position, -1 = masked).  Project-then-gather is the same
    arithmetic as the reference's gather-then-project (the projections have no bias) at 1/25 of its k / v projection flops and without
    the [L, 25, C] gathered tensors; forward = K5, backward = gf_window_cross_attention_backward."""

    @staticmethod
    def forward(ctx, q, kmap, vmap, win, nhead):
        out = ops.window_cross_attention(q, kmap, vmap, win, None, nhead)
        ctx.save_for_backward(q, kmap, vmap, win)
        ctx.nhead = nhead
        return out

    @staticmethod
    def backward(ctx, dout):
        q, kmap, vmap, win = ctx.saved_tensors
        if _GATHER_K5_BACKWARD[0]:
            # round 6: the cells' sums gathered along the inverse of the window table (one sort per table, kept on the table: both
            # 'cross' layers of a step use the same one) instead of fp32 atomic adds: bit-reproducible, no fp32 maps
            index = getattr(win, '_gf_inverse_index', None)
            if index is None or index[2] != (kmap.shape[1], win._version):
                index = ops.window_inverse_index(win, kmap.shape[1]) + ((kmap.shape[1], win._version),)
                win._gf_inverse_index = index
            dq, dk, dv = ops.window_cross_attention_backward_gather(q, kmap, vmap, dout, win, index[:2], ctx.nhead)
            return dq, dk, dv, None, None
        dq, dk, dv = ops.window_cross_attention_backward(q, kmap, vmap, dout, win, ctx.nhead)
        return dq, dk.to(kmap.dtype), dv.to(vmap.dtype), None, None


class HipFineMatch(torch.autograd.Function):
    """FineMatching2.forward + get_fine_match (model/fine_matching2.py:21-126) = K8: returns (fine_matrix [M, 25, 25] fp32, m_bids, mkpts0_f,
    mkpts1_f, mconf); only fine_matrix carries a gradient (to the two window tensors), through gf_fine_match_backward."""

    @staticmethod
    def forward(ctx, f0, f1, temperature, thr, b_ids, mk0c, mk1c, cscale, c2f, fscale, scale0, scale1):
        out = ops.fine_match(f0, f1, temperature, thr, b_ids, mk0c, mk1c, cscale, c2f, fscale, scale0, scale1)
        n = int(out['count'][0])
        ctx.temperature = temperature
        ctx.save_for_backward(f0, f1)
        res = (out['fine_matrix'], out['m_bids'][:n], out['mkpts0_f'][:n], out['mkpts1_f'][:n], out['mconf'][:n])
        ctx.mark_non_differentiable(*res[1:])
        return res

    @staticmethod
    def backward(ctx, dconf, *unused):
        f0, f1 = ctx.saved_tensors
        df0, df1 = ops.fine_match_backward(f0, f1, ctx.temperature, dconf)
        return (df0, df1) + (None,) * 10


def _pad_width(c):
    """the channel widths K10 is built for: the pyramid's 196-channel level runs zero-padded to 224 (model/backbone.py)"""
    return {128: 128, 196: 224, 224: 224, 256: 256}.get(int(c))


def _pad_channels(x, cp):
    """channels_last [N, C, H, W] -> channels_last [N, cp, H, W], the new channels zero"""
    if x.shape[1] == cp:
        return x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)
    out = torch.empty((x.shape[0], cp, x.shape[2], x.shape[3]), dtype=x.dtype, device=x.device, memory_format=torch.channels_last).zero_()
    out[:, :x.shape[1]] = x
    return out


def _conv_stream(w16, cout_p, cin_p):
    """K10's fragment stream of the zero-padded [cout_p, cin_p, 3, 3] form of w16 (packed on the device: the weights move every step)"""
    from .. import fused
    co, ci = w16.shape[:2]
    if (co, ci) != (cout_p, cin_p):
        wp = torch.zeros((cout_p, cin_p, 3, 3), dtype=w16.dtype, device=w16.device)
        wp[:co, :ci] = w16
        w16 = wp
    return fused.pack_conv3x3_stream(w16.contiguous())


def conv3x3_supported(x, w, stride=1):
    """a bias-free 3x3 / stride 1 / pad 1 convolution of a 16-bit channels_last map whose widths K10 has kernels for - forward AND,
    with the widths swapped, backward-data"""
    from .. import fused
    if not (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float16, torch.bfloat16) and tuple(w.shape[2:]) == (3, 3) and stride == 1):
        return False
    cip, cop = _pad_width(w.shape[1]), _pad_width(w.shape[0])
    return cip is not None and cop is not None and fused.conv3x3_supported(cip, cop) and fused.conv3x3_supported(cop, cip)


_OWN_CONV_WGRAD = [True]      # the weight gradients of HipConv3x3 on csrc/k10_conv3x3_wgrad.hip (False: aten::convolution_backward, for A/B)


class HipConv3x3(torch.autograd.Function):
    """y = conv2d(x, w, stride 1, padding 1), no bias (the backbone's BasicBlock / FPN-head convolutions, resnet_fpn.py:9-40,60-83): forward on
    K10 (`gf_conv3x3_nhwc`, no epilogue - train-mode BatchNorm follows as its own op), backward-DATA on K10 as well - dX = conv(dY, w'),
    w'[ci, co, ky, kx] = w[co, ci, 2 - ky, 2 - kx]: the same kernel on dY with the transposed, flipped weights - and backward-WEIGHTS on
    K10's weight-gradient kernel (`gf_conv3x3_wgrad_nhwc`, round 6: pixels as the contraction index, bit-reproducible).  x: 16-bit
    channels_last; w: fp32 master or 16-bit [cout, cin, 3, 3]; 196-channel operands are zero-padded to 224 on the way in (the padded
    x is what is kept for the backward) and sliced on the way out (one copy each)."""

    @staticmethod
    def forward(ctx, x, w):
        from .. import fused
        w16 = w if w.dtype == x.dtype else WEIGHTS.cast(w, x.dtype)
        co, ci = w16.shape[:2]
        cop, cip = _pad_width(co), _pad_width(ci)
        xp = _pad_channels(x, cip)
        y = fused.conv3x3(xp, _conv_stream(w16, cop, cip), cop)
        if cop != co:
            y = y[:, :co].contiguous(memory_format=torch.channels_last)
        ctx.wdtype = w.dtype
        ctx.save_for_backward(xp, w16)
        return y

    @staticmethod
    def backward(ctx, dy):
        from .. import fused
        xp, w16 = ctx.saved_tensors
        co, ci = w16.shape[:2]
        cop, cip = _pad_width(co), _pad_width(ci)
        dx = dw = None
        dyp = _pad_channels(dy, cop)
        if ctx.needs_input_grad[0]:
            wt = w16.flip(2, 3).transpose(0, 1)                            # [cin, cout, 3, 3]
            dx = fused.conv3x3(dyp, _conv_stream(wt, cip, cop), cip)
            if cip != ci:
                dx = dx[:, :ci].contiguous(memory_format=torch.channels_last)
        if ctx.needs_input_grad[1]:
            if _OWN_CONV_WGRAD[0]:
                dw = fused.conv3x3_wgrad(xp, dyp, ci, co)
            else:
                dw = torch.ops.aten.convolution_backward(dyp[:, :co], xp[:, :ci], w16, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                         [False, True, False])[1]
            dw = dw.to(ctx.wdtype)
        return dx, dw


def _pad32(c):
    return (int(c) + 31) // 32 * 32


class HipConv1x1(torch.autograd.Function):
    """y = conv2d(x, w) for a 1x1 / stride-1 kernel without bias (the FPN's lateral and output convolutions, resnet_fpn.py:69-83) on the K3
    engine: forward `gf_conv1x1_nhwc`, backward-data the same kernel with the transposed weight, backward-weights `gf_linear_wgrad` on the
    pixel rows (round 6; the library's 1x1 at 16 x 128 -> 196 x 320 x 320 took 1.97 ms forward: 42 TFLOP/s).  x: 16-bit channels_last; w:
    fp32 master or 16-bit [cout, cin, 1, 1]; widths that are not multiples of 32 (196) are zero-padded on the way in and sliced on the way
    out (one copy each)."""

    @staticmethod
    def forward(ctx, x, w):
        w16 = w if w.dtype == x.dtype else WEIGHTS.cast(w, x.dtype)
        co, ci = w16.shape[:2]
        cop, cip = _pad32(co), _pad32(ci)
        wp = w16.reshape(co, ci)
        if (cop, cip) != (co, ci):
            wp = torch.zeros(cop, cip, dtype=w16.dtype, device=w16.device)
            wp[:co, :ci] = w16.reshape(co, ci)
        xp = _pad_channels(x, cip)
        y = ops.conv1x1(xp, wp)
        if cop != co:
            y = y[:, :co].contiguous(memory_format=torch.channels_last)
        ctx.wdtype, ctx.dims = w.dtype, (co, ci)
        ctx.save_for_backward(xp, wp)
        return y

    @staticmethod
    def backward(ctx, dy):
        xp, wp = ctx.saved_tensors
        co, ci = ctx.dims
        cop, cip = wp.shape
        dx = dw = None
        dyp = _pad_channels(dy, cop)
        if ctx.needs_input_grad[0]:
            dx = ops.conv1x1(dyp, wp.t().contiguous())
            if cip != ci:
                dx = dx[:, :ci].contiguous(memory_format=torch.channels_last)
        if ctx.needs_input_grad[1]:
            rows = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])          # channels_last maps as [pixels, C] rows (a view)
            dw = ops.linear_wgrad(rows(dyp), rows(xp))[:co, :ci].reshape(co, ci, 1, 1).to(ctx.wdtype)
        return dx, dw


def conv1x1(x, w):
    return HipConv1x1.apply(x, w)


class HipUpsampleBilinear(torch.autograd.Function):
    """F.interpolate(x, size, mode='bilinear', align_corners=True) of the FPN's top-down merge (resnet_fpn.py:104-105, :110-111): the forward
    is the library's, the backward `gf_upsample_bilinear_backward_nhwc` - a gather per low-resolution pixel where the library scatters
    with atomics (3.3 ms per call at 16 x 196 x 320 x 320, 0.4 here)."""

    @staticmethod
    def forward(ctx, x, size):
        ctx.hw = tuple(x.shape[2:])
        return torch.nn.functional.interpolate(x, size=size, mode='bilinear', align_corners=True)

    @staticmethod
    def backward(ctx, dy):
        dy = dy if dy.is_contiguous(memory_format=torch.channels_last) else dy.contiguous(memory_format=torch.channels_last)
        return ops.upsample_bilinear_backward(dy, *ctx.hw), None


def upsample_bilinear(x, size):
    return HipUpsampleBilinear.apply(x, size)


def conv3x3(x, w):
    return HipConv3x3.apply(x, w)


def linear_attention(q, k, v, nhead, q_mask=None, kv_mask=None):
    return HipLinearAttention.apply(q, k, v, nhead, q_mask, kv_mask)


def window_linear_attention(q, k, v, nhead):
    return HipWindowLinearAttention.apply(q, k, v, nhead)


def full_attention(q, k, v, nhead=4):
    return HipFullAttention.apply(q, k, v, nhead)


def window_cross_attention(q, kmap, vmap, win, nhead=4):
    return HipWindowCrossAttention.apply(q, kmap, vmap, win, nhead)


def linear(x, w, x2=None, act=None):
    return HipLinear.apply(x, w, x2, act)


def layer_norm(y, gamma, beta, eps=1e-5):
    return HipLayerNorm.apply(y, gamma, beta, eps)


def supported(x, *widths):
    """The HIP chain is built for 16-bit CUDA activations and channel counts that are multiples of 128."""
    return x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and all(w % 128 == 0 for w in widths)
