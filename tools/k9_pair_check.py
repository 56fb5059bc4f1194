"""Round-6 development check of the wave-pair encoder kernels (csrc/k9_encoder_pair.hip) against the four-wave kernels of rounds 2-5:
   GF_K9_PAIR=0 python tools/k9_pair_check.py save ; GF_K9_PAIR=1 python tools/k9_pair_check.py save ; python tools/k9_pair_check.py cmp
(the two forms consume different weight-stream orders, so each runs in a process of its own)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import torch

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')


def run():
    import geoformer_oracle as O
    from geoformer_amd import fused
    from geoformer_amd.model.modules import LoFTREncoderLayer
    res = {}
    W = O.make_weights()
    for dt in (torch.float16, torch.bfloat16):
        for (N, L) in ((2, 300), (4, 6400)):
            pfx = 'loftr_coarse.layers.0.'
            layer = LoFTREncoderLayer(256, 8, 'linear', 'relu')
            layer.load_state_dict({k[len(pfx):]: v for k, v in W.items() if k.startswith(pfx)})
            layer = layer.cuda()
            g = torch.Generator(device='cuda').manual_seed(11)
            x = (torch.randn(N, L, 256, device='cuda', generator=g) * 0.7).to(dt)
            qm = torch.rand(N, L, device='cuda', generator=g) > 0.05
            w = layer.weights(dt)
            key = f'{str(dt)[6:]}_{N}x{L}'
            state = fused.encoder_kv_state(x, w['stream_kv'])
            res[key + '_state'] = state.cpu()
            res[key + '_layer'] = fused.encoder_layer(x, w['stream'], w['ln'], 1e-5, 1e-5, 0, kv_state=state, source_len=L).float().cpu()
            res[key + '_masked'] = fused.encoder_layer(x, w['stream'], w['ln'], 1e-5, 1e-5, 0, kv_state=state, source_len=L, q_mask=qm).float().cpu()
            yt, st2 = fused.encoder_layer(x, w['stream'], w['ln'], 1e-5, 1e-5, 0, kv_state=state, source_len=L, tail_stream=w['stream_kv'], tail_first=0)
            res[key + '_tail_out'] = yt.float().cpu()
            res[key + '_tail_state'] = st2.cpu()
            res[key + '_tail_state_sep'] = fused.encoder_kv_state(yt, w['stream_kv']).cpu()
            res[key + '_finish'] = fused.encoder_layer(x, w['stream_finish'], w['ln'], 1e-5, 1e-5, 1, msg=x).float().cpu()
            torch.cuda.synchronize()
            print(key, 'done', flush=True)
    return res


def main():
    mode = sys.argv[1]
    os.makedirs(OUT, exist_ok=True)
    if mode == 'save':
        tag = os.environ.get('GF_K9_PAIR', '1')
        torch.save(run(), os.path.join(OUT, f'k9_pair_{tag}.pt'))
        return
    a, b = torch.load(os.path.join(OUT, 'k9_pair_0.pt')), torch.load(os.path.join(OUT, 'k9_pair_1.pt'))
    bad = 0
    for k in a:
        d = (a[k] - b[k]).abs()
        scale = float(a[k].abs().max())
        nanb = int(torch.isnan(b[k]).sum())
        line = f'{k:34s} max|old| {scale:9.3f}  max diff {float(d.max()):.3e}  mean diff {float(d.mean()):.3e}  nan(new) {nanb}'
        if d.dim() == 3 and float(d.max()) > 0.05 * max(scale, 1e-6):
            bad += 1
            # localise: by channel half / token group / tile
            N, L, C = d.shape
            line += '\n    by channel 32-tile: ' + ' '.join(f'{float(d[..., 32 * t:32 * t + 32].max()):.2e}' for t in range(8))
            Lp = (L // 128) * 128
            if Lp:
                g = d[:, :Lp].reshape(N, Lp // 128, 4, 32, C)
                line += '\n    by token group    : ' + ' '.join(f'{float(g[:, :, i].max()):.2e}' for i in range(4))
        print(line)
    for k in b:
        if k.endswith('_tail_state'):
            same = torch.equal(b[k], b[k + '_sep'])
            print(f'{k}: tail state == separate pass (new): {same}   max diff {float((b[k] - b[k + "_sep"]).abs().max()):.3e}')
            same_out = torch.equal(b[k.replace('_tail_state', '_tail_out')], b[k.replace('_tail_state', '_layer')])
            print(f'{k}: tail launch output == plain launch output: {same_out}')
    print('LARGE DIFFERENCES:', bad)


if __name__ == '__main__':
    main()
