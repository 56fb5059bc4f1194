"""The fused encoder-layer kernels (csrc/k9_encoder_fused.hip, geoformer_amd/fused.py).

CPU: the two weight-fragment orders of fused.fragments against their definition.
GPU: gf_encoder_kv_state / gf_encoder_layer against the oracle's 'fused' storage mode (the reference's arithmetic
with a round trip through fp16 at the kernels' rounding points) on the same inputs: kv state to 1e-3, layer outputs
equal to two fp16 ulp (see _ulp_close)."""
import numpy as np
import pytest
import torch

import geoformer_oracle as O

DEV = 'cuda:0'
PFX = 'loftr_coarse.layers.2.'
GPFX = 'geo_module.des_transformer.layers.1.'


def test_fragment_orders():
    from geoformer_amd.fused import fragments, pack_kv_stream, pack_layer_stream
    g = torch.Generator().manual_seed(3)
    w = torch.randn(64, 96, generator=g)
    fs, fp = fragments(w, 'std'), fragments(w, 'perm')
    assert fs.shape == (2, 6, 64, 8) and fp.shape == (2, 6, 64, 8)
    for nb in range(2):
        for ks in range(6):
            for lane in (0, 5, 31, 32, 47, 63):
                for j in range(8):
                    r, h = lane & 31, lane >> 5
                    assert fs[nb, ks, lane, j] == w[32 * nb + r, 16 * ks + 8 * h + j]
                    t, s = ks >> 1, ks & 1
                    assert fp[nb, ks, lane, j] == w[32 * nb + r, 32 * t + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)]
    c = 256
    wq, wm, w1, w2 = (torch.randn(c, c, generator=g), torch.randn(c, c, generator=g), torch.randn(2 * c, 2 * c, generator=g),
                      torch.randn(c, 2 * c, generator=g))
    assert pack_layer_stream(wq, wm, w1, w2).numel() == 32 * 32 * 64 * 8          # 32 blocks of 32 fragments: 1 MiB in fp16
    assert pack_layer_stream(None, wm, w1, w2).numel() == 28 * 32 * 64 * 8
    assert pack_kv_stream(wq, wm).numel() == 8 * 32 * 64 * 8
    # every weight element appears exactly once in the stream
    s = pack_layer_stream(wq, wm, w1, w2)
    assert torch.equal(torch.sort(s)[0], torch.sort(torch.cat([wq.flatten(), wm.flatten(), w1.flatten(), w2.flatten()]))[0])


def _layer(prefix, attention, activation, nhead):
    from geoformer_amd.model.modules import LoFTREncoderLayer
    W = O.make_weights()
    m = LoFTREncoderLayer(256, nhead, attention, activation)
    m.load_state_dict({k[len(prefix):]: v for k, v in W.items() if k.startswith(prefix)})
    return m.to(DEV), W


def _ulp_close(got, want, what, st=torch.float16):
    """fp16 outputs of a chain of five rounded GEMM stages: the two sides sum in different orders, so a value that sits
    within ~1e-7 of a rounding boundary lands on the other fp16 neighbour, and every such flip perturbs the whole
    next layer by a fraction of an ulp - exact equality is not attainable; agreement is to 2 ulp, with a mean
    difference far below one ulp."""
    got, want = got.float().cpu(), want.float()
    k = 1 if st == torch.float16 else 8                      # bf16: 8 significant bits against 11
    torch.testing.assert_close(got, want, rtol=4e-3 * k, atol=4e-3 * k, msg=lambda m: f'{what}: {m}')
    assert float((got - want).abs().mean()) < 3e-4 * k, what


@pytest.mark.gpu
@pytest.mark.parametrize('st', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('masked', [False, True])
def test_kv_state_vs_oracle(masked, st):
    from geoformer_amd import fused
    layer, W = _layer(PFX, 'linear', 'relu', 8)
    g = torch.Generator().manual_seed(5)
    N, S = 2, 300                                           # 3 tiles per image, the last one ragged
    src = O.rt(torch.randn(N, S, 256, generator=g) * 0.8, st)
    km = None
    if masked:
        km = torch.ones(N, S, dtype=torch.bool); km[0, 250:] = False; km[1, 10:40] = False
    k = torch.nn.functional.linear(src, O.rt(W[PFX + 'k_proj.weight'], st)).view(N, S, 8, 32)
    v = torch.nn.functional.linear(src, O.rt(W[PFX + 'v_proj.weight'], st)).view(N, S, 8, 32)
    Kf = O._phi(k)
    if masked:
        Kf = Kf * km[:, :, None, None]
    K, V = O.rt(Kf, st), O.rt(v, st)
    KV = torch.einsum('nshd,nshv->nhdv', K, V).reshape(N, 256, 32)          # [c = h*32 + d][v]
    ref = torch.cat([KV.reshape(N, -1), Kf.sum(1).reshape(N, 256)], 1)
    w = layer.weights(st)
    got = fused.encoder_kv_state(src.to(DEV).to(st), w['stream_kv'], None if km is None else km.to(DEV)).cpu()
    tol = 1e-3 if st == torch.float16 else 8e-3
    torch.testing.assert_close(got, ref, rtol=tol, atol=tol * float(ref.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize('st', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('L,S,masked', [(256, 256, False), (300, 200, False), (200, 300, True)])
def test_linear_attention_layer_vs_oracle(L, S, masked, st):
    layer, W = _layer(PFX, 'linear', 'relu', 8)
    g = torch.Generator().manual_seed(7)
    N = 2
    x = O.rt(torch.randn(N, L, 256, generator=g) * 0.7, st)
    src = O.rt(torch.randn(N, S, 256, generator=g) * 0.7, st)
    xm = sm = None
    if masked:
        xm = torch.ones(N, L, dtype=torch.bool); xm[1, 150:] = False
        sm = torch.ones(N, S, dtype=torch.bool); sm[0, 280:] = False; sm[1, :17] = False
    ref = O.encoder_layer_fused(W, PFX, x, src, 8, st, xm, sm)
    with torch.no_grad():
        got = layer(x.to(DEV).to(st), src.to(DEV).to(st), None if xm is None else xm.to(DEV), None if sm is None else sm.to(DEV))
    assert got.dtype == st
    _ulp_close(got, ref, f'layer L={L} S={S}', st)


@pytest.mark.gpu
@pytest.mark.parametrize('st', [torch.float16, torch.bfloat16])
def test_finish_vs_oracle_tanh_and_skip_flags(st):
    """The Geo form: attention output given, Tanh MLP, per-sample 'layer skipped' predicate."""
    layer, W = _layer(GPFX, 'full', 'tanh', 4)
    g = torch.Generator().manual_seed(9)
    N, L = 3, 200
    x = O.rt(torch.randn(N, L, 256, generator=g) * 0.7, st)
    msg = O.rt(torch.randn(N, L, 256, generator=g) * 0.5, st)
    flag = torch.tensor([1, 0, 5], dtype=torch.int32)
    ref = O._finish_fused(W, GPFX, x, msg, 'geo', st)
    ref[1] = x[1]
    with torch.no_grad():
        got = layer.finish(x.to(DEV).to(st), msg.to(DEV).to(st), flag.to(DEV), L)
    _ulp_close(got, ref, 'finish tanh', st)
    assert torch.equal(got[1].cpu(), x[1].to(st))           # a skipped sample is copied bit for bit


@pytest.mark.gpu
def test_full_size_layer_against_unfused_chain():
    """L = S = 6400 (the 640x640 grid), 4 images: the fused layer against the K3 + K2 kernel chain it replaces (both
    fp16 storage; they round at different points, so agreement is to fp16 resolution, not bitwise)."""
    from geoformer_amd import ops
    layer, _ = _layer(PFX, 'linear', 'relu', 8)
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(4, 6400, 256, generator=g) * 0.7).to(DEV).half()
    src = (torch.randn(4, 6400, 256, generator=g) * 0.7).to(DEV).half()
    with torch.no_grad():
        got = layer(x, src)
        w = layer.weights(torch.float16)
        q = ops.linear(x, w['q'])
        kv = ops.linear(src, w['kv'])
        msg = ops.linear_attention(q, kv[..., :256], kv[..., 256:], 8)
        msg = ops.linear(msg, w['merge'], epilogue=ops.EPI_LN, ln=w['n1'], eps=layer.norm1.eps)
        hid = ops.linear(x, w['w1'], a2=msg, epilogue=ops.EPI_RELU)
        want = ops.linear(hid, w['w2'], epilogue=ops.EPI_LN_RES, ln=w['n2'], eps=layer.norm2.eps, residual=x)
    torch.testing.assert_close(got.float(), want.float(), rtol=1e-2, atol=1e-2)
    assert float((got.float() - want.float()).abs().mean()) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize('st', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('L,masked,tail_first', [(256, False, 0), (300, True, 1), (6400, False, 2)])
def test_state_tail_equals_a_separate_state_pass(L, masked, tail_first, st):
    """gf_encoder_layer_kv (round 4): the layer's output is unchanged by the tail, and the states the launch leaves for the
    images tail_first.. are BIT-IDENTICAL to gf_encoder_kv_state run on that output with the consumer's weights (same tile,
    same operand order, same reduction order) - ragged last tile, query masks as the source masks, tails on a suffix of the batch."""
    from geoformer_amd import fused
    layer, _ = _layer(PFX, 'linear', 'relu', 8)
    consumer, _ = _layer('loftr_coarse.layers.3.', 'linear', 'relu', 8)
    g = torch.Generator().manual_seed(13)
    N = 3
    x = (torch.randn(N, L, 256, generator=g) * 0.7).to(DEV).to(st)
    src = (torch.randn(N, L, 256, generator=g) * 0.7).to(DEV).to(st)
    xm = None
    if masked:
        xm = torch.ones(N, L, dtype=torch.bool); xm[1, 150:] = False; xm[2, :33] = False
        xm = xm.to(DEV)
    with torch.no_grad():
        state = layer.kv_state(src)
        plain = layer.forward_state(x, state, L, xm)
        out, tail = layer.forward_state(x, state, L, xm, tail_layer=consumer, tail_first=tail_first)
        want = consumer.kv_state(plain[tail_first:], None if xm is None else xm[tail_first:])
    assert torch.equal(out, plain)
    assert tail.shape == (N - tail_first, 256 * 32 + 256)
    assert torch.equal(tail, want)
    # into a row range of a larger buffer (how LocalFeatureTransformer hands a 'self' layer its two halves)
    buf = torch.full((N + 2, 256 * 32 + 256), -7.0, device=DEV)
    with torch.no_grad():
        layer.forward_state(x, state, L, xm, tail_layer=consumer, tail_first=tail_first, tail_out=buf[1:1 + N - tail_first])
    assert torch.equal(buf[1:1 + N - tail_first], want) and float(buf[0, 0]) == -7.0 and float(buf[-1, -1]) == -7.0


@pytest.mark.gpu
@pytest.mark.parametrize('masked', [False, True])
def test_transformer_with_state_tails_equals_layer_by_layer(masked):
    """LocalFeatureTransformer on the fused path (states handed from producer to consumer through the tails) against the same
    eight layers run one by one, each with its own gf_encoder_kv_state pass (round 3's schedule): bit-identical features."""
    from geoformer_amd.model.modules import LocalFeatureTransformer
    from geoformer_amd.model.cvpr_ds_config import get_default_cfg
    cfg = get_default_cfg()['coarse']
    m = LocalFeatureTransformer(cfg)
    W = O.make_weights()
    m.load_state_dict({k[len('loftr_coarse.'):]: v for k, v in W.items() if k.startswith('loftr_coarse.')})
    m = m.to(DEV)
    g = torch.Generator().manual_seed(17)
    n, L = 2, 420
    f0 = (torch.randn(n, L, 256, generator=g) * 0.7).to(DEV).half()
    f1 = (torch.randn(n, L, 256, generator=g) * 0.7).to(DEV).half()
    m0 = m1 = None
    if masked:
        m0 = torch.ones(n, L, dtype=torch.bool, device=DEV); m0[0, 400:] = False
        m1 = torch.ones(n, L, dtype=torch.bool, device=DEV); m1[1, :50] = False
    with torch.no_grad():
        a0, a1 = m(f0, f1, m0, m1)
        b0, b1 = f0, f1
        for layer, name in zip(m.layers, m.layer_names):
            if name == 'self':
                b0, b1 = layer(b0, b0, m0, m0), layer(b1, b1, m1, m1)
            else:
                b0 = layer(b0, b1, m0, m1)
                b1 = layer(b1, b0, m1, m0)
    assert torch.equal(a0, b0) and torch.equal(a1, b1)


@pytest.mark.gpu
@pytest.mark.parametrize('masked', [False, True])
def test_transformer_unequal_pair_with_state_tails_equals_layer_by_layer(masked):
    """Round 6: a pair whose images have different token counts (the HPatches loop's shapes, BASELINE configs[1]) on the fused path with state
    tails (5 state passes of their own instead of 16) against the same eight layers run call by call, each with its own gf_encoder_kv_state
    pass: bit-identical features."""
    from geoformer_amd.model.modules import LocalFeatureTransformer
    from geoformer_amd.model.cvpr_ds_config import get_default_cfg
    cfg = get_default_cfg()['coarse']
    m = LocalFeatureTransformer(cfg)
    W = O.make_weights()
    m.load_state_dict({k[len('loftr_coarse.'):]: v for k, v in W.items() if k.startswith('loftr_coarse.')})
    m = m.to(DEV)
    g = torch.Generator().manual_seed(19)
    n, L0, L1 = 1, 4800, 4560
    f0 = (torch.randn(n, L0, 256, generator=g) * 0.7).to(DEV).to(torch.bfloat16)
    f1 = (torch.randn(n, L1, 256, generator=g) * 0.7).to(DEV).to(torch.bfloat16)
    m0 = m1 = None
    if masked:
        m0 = torch.ones(n, L0, dtype=torch.bool, device=DEV); m0[0, 4700:] = False
        m1 = torch.ones(n, L1, dtype=torch.bool, device=DEV); m1[0, :50] = False
    with torch.no_grad():
        a0, a1 = m(f0, f1, m0, m1)
        b0, b1 = f0, f1
        for layer, name in zip(m.layers, m.layer_names):
            if name == 'self':
                b0, b1 = layer(b0, b0, m0, m0), layer(b1, b1, m1, m1)
            else:
                b0 = layer(b0, b1, m0, m1)
                b1 = layer(b1, b0, m1, m0)
    assert a0.shape == (n, L0, 256) and a1.shape == (n, L1, 256)
    assert torch.equal(a0, b0) and torch.equal(a1, b1)
