"""K11, the fused fine-level encoder layer (csrc/k11_fine_layer.hip, geoformer_amd/fused.py:pack_fine_layer_stream).

CPU: every weight element appears exactly once in the 320-KiB stream, in the documented step order.
GPU: gf_fine_layer against the oracle's 'chain' storage mode (encoder_layer_chain + linear_attention_window: the reference's
LoFTREncoderLayer / LinearAttention arithmetic, loftr_module/transformer.py:37-60 and linear_attention.py:21-51, with round
trips through the storage type at the kernel's rounding points) on the same windows: equal to two ulp of the storage type;
and against the K3 / K2 launch chain it replaces."""
import numpy as np
import pytest
import torch

import geoformer_oracle as O

DEV = 'cuda:0'
PFX = 'loftr_fine.layers.1.'


def test_fine_stream_order():
    from geoformer_amd.fused import fragments, pack_fine_layer_stream
    g = torch.Generator().manual_seed(7)
    c = 128
    wq, wk, wv, wm = (torch.randn(c, c, generator=g) for _ in range(4))
    w1, w2 = torch.randn(2 * c, 2 * c, generator=g), torch.randn(c, 2 * c, generator=g)
    s = pack_fine_layer_stream(wq, wk, wv, wm, w1, w2)
    assert s.numel() == 10 * 32 * 64 * 8                                              # 10 blocks of 32 fragments
    allw = torch.cat([t.flatten() for t in (wq, wk, wv, wm, w1, w2)])
    assert torch.equal(torch.sort(s)[0], torch.sort(allw)[0])                         # each element exactly once
    f = s.view(80, 4, 64, 8)                                                          # [step][fragment][lane][element]
    fk, fv, fq, fm = fragments(wk, 'std'), fragments(wv, 'std'), fragments(wq, 'std'), fragments(wm, 'perm')
    for nb in range(4):                                                               # steps 4nb .. 4nb+3: W_k[nb] halves, W_v[nb] halves
        for hf in range(2):
            for i in range(4):
                assert torch.equal(f[4 * nb + hf, i], fk[nb, 4 * hf + i]) and torch.equal(f[4 * nb + 2 + hf, i], fv[nb, 4 * hf + i])
                assert torch.equal(f[16 + 2 * nb + hf, i], fq[nb, 4 * hf + i]) and torch.equal(f[24 + 2 * nb + hf, i], fm[nb, 4 * hf + i])
    f1x, f1m, f2 = fragments(w1[:, :c], 'std'), fragments(w1[:, c:], 'perm'), fragments(w2, 'perm')
    for hb in range(8):
        base = 32 + 6 * hb
        for hf in range(2):
            for i in range(4):
                assert torch.equal(f[base + hf, i], f1x[hb, 4 * hf + i]) and torch.equal(f[base + 2 + hf, i], f1m[hb, 4 * hf + i])
        for sx in range(2):
            for nb in range(4):
                assert torch.equal(f[base + 4 + sx, nb], f2[nb, 2 * hb + sx])


def _layer(prefix=PFX):
    from geoformer_amd.model.modules import LoFTREncoderLayer
    W = O.make_weights()
    m = LoFTREncoderLayer(128, 8, 'linear', 'relu')
    m.load_state_dict({k[len(prefix):]: v for k, v in W.items() if k.startswith(prefix)})
    return m.to(DEV), W


def _ulp_close(got, want, what, st, layers=1):
    """Two ulp of the storage type per element, mean difference far below one ulp (per layer: a value that lands on the
    other side of a rounding boundary perturbs the whole next layer by a fraction of an ulp)."""
    got, want = got.float().cpu(), want.float()
    k = (1 if st == torch.float16 else 8) * layers
    torch.testing.assert_close(got, want, rtol=4e-3 * k, atol=4e-3 * k, msg=lambda m: f'{what}: {m}')
    assert float((got - want).abs().mean()) < 3e-4 * k, (what, float((got - want).abs().mean()))


@pytest.mark.gpu
@pytest.mark.parametrize('st', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('Nw,Lw,cross', [(8, 25, False), (37, 25, True), (3, 25, False), (2051, 25, True), (5, 32, False), (9, 17, True)])
def test_fine_layer_vs_chain_oracle(st, Nw, Lw, cross):
    """Self (src = x) and cross (src != x) forms; window counts that leave spare waves in the last group (3, 37), more groups
    than workgroups (2051 windows = 257 groups on 256 workgroups: the ring runs on across groups), full 32-token windows and
    short ones (masked token slots)."""
    layer, W = _layer()
    g = torch.Generator().manual_seed(100 + Nw + Lw)
    x = O.rt(torch.randn(Nw, Lw, 128, generator=g) * 0.8, st)
    src = O.rt(torch.randn(Nw, Lw, 128, generator=g) * 0.8, st) if cross else x
    ref = O.encoder_layer_chain(W, PFX, x, src, 8, st)
    xd = x.to(DEV).to(st)
    sd = src.to(DEV).to(st) if cross else xd
    got = layer(xd, sd)
    assert got.dtype == st and got.shape == (Nw, Lw, 128)
    _ulp_close(got, ref, f'fine layer {Nw}x{Lw} cross={cross}', st)
    # the launch chain it replaces (K3 x 5 + K2) agrees to the same resolution
    from geoformer_amd import ops
    w = layer.weights(st)
    q = ops.linear(xd, w['q']); kv = ops.linear(sd, w['kv'])
    msg = ops.linear_attention(q, kv[..., :128], kv[..., 128:], 8)
    chain = layer.finish(xd, msg)
    _ulp_close(got, chain.float().cpu(), 'against the K3/K2 chain', st)


@pytest.mark.gpu
def test_fine_layer_repeatable_and_in_transformer():
    """25 identical launches at the nominal-load size (37 k windows) give bit-identical outputs (LDS-DMA ring + counted waits:
    a race shows up as rare differing tiles), also with a second stream streaming beside it; and LocalFeatureTransformer's
    fine configuration (self, cross) through the fused layers equals the oracle's chain mode."""
    from geoformer_amd.model.modules import LocalFeatureTransformer
    st = torch.float16
    layer, W = _layer()
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(37120, 25, 128, generator=g) * 0.8).to(st).to(DEV)
    first = layer(x, x).clone()
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, device=DEV, dtype=torch.float32)
    for it in range(25):
        if it % 2:
            with torch.cuda.stream(side):
                junk.normal_()                                   # an HBM-streaming neighbour on another stream
        assert torch.equal(layer(x, x), first), it
    torch.cuda.synchronize()
    tr = LocalFeatureTransformer({'d_model': 128, 'nhead': 8, 'layer_names': ['self', 'cross'], 'attention': 'linear'})
    tr.load_state_dict({k[len('loftr_fine.'):]: v for k, v in W.items() if k.startswith('loftr_fine.')})
    tr = tr.to(DEV)
    f0 = O.rt(torch.randn(41, 25, 128, generator=g), st)
    f1 = O.rt(torch.randn(41, 25, 128, generator=g), st)
    o0, o1 = tr(f0.to(DEV).to(st), f1.to(DEV).to(st))
    r0 = O.encoder_layer_chain(W, 'loftr_fine.layers.0.', f0, f0, 8, st)
    r1 = O.encoder_layer_chain(W, 'loftr_fine.layers.0.', f1, f1, 8, st)
    r0 = O.encoder_layer_chain(W, 'loftr_fine.layers.1.', r0, r1, 8, st)
    r1 = O.encoder_layer_chain(W, 'loftr_fine.layers.1.', r1, r0, 8, st)
    _ulp_close(o0, r0, 'loftr_fine f0', st, layers=2)
    _ulp_close(o1, r1, 'loftr_fine f1', st, layers=2)
