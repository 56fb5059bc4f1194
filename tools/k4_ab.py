"""K4 forms side by side on one box: python tools/k4_ab.py [keys=1195] [QB,WV[,FORM] | def,- ...] - every (GF_K4_QB, GF_K4_WV[, GF_K4_FORM])
variant in a process of its own (the switches are read once), time per 16-image call and a digest of the output (the forms run the
same arithmetic per query: the digests must be equal)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, hashlib
sys.path.insert(0, %r)
import torch
from geoformer_amd import ops
K = int(sys.argv[1])
N, L = int(os.environ.get('K4AB_N', '16')), int(os.environ.get('K4AB_L', '6400'))       # (K4AB_N / K4AB_L: other batch shapes)
g = torch.Generator(device='cuda').manual_seed(5)
q = torch.randn(N, L, 256, device='cuda', generator=g).half()
kv = torch.randn(N, L, 512, device='cuda', generator=g).half()
idx = torch.stack([torch.randperm(L, device='cuda', generator=g).sort()[0] for _ in range(N)]).int()
nk = torch.full((N,), K, device='cuda', dtype=torch.int32)
f = lambda: ops.self_attention_gathered(q, kv[..., :256], kv[..., 256:], idx, nk)
for _ in range(3):
    out = f()
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        f()
    b.record(); torch.cuda.synchronize()
    best = min(best, a.elapsed_time(b) / 20 * 1e3)
dig = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:12]
print(f'{best:.1f} us  {4.0 * N * L * K * 256 / best / 1e6:.0f} TFLOP/s  digest {dig}')
''' % ROOT

K = sys.argv[1] if len(sys.argv) > 1 else '1195'
variants = [v.split(',') for v in (sys.argv[2:] or ['2,4', '1,4', '2,8', '1,8'])]
for v in variants:
    env = dict(os.environ, GF_K4_QB=v[0], GF_K4_WV=v[1]) if v[0] != 'def' else dict(os.environ)      # 'def,-': the library's own choice
    if len(v) > 2:
        env['GF_K4_FORM'] = v[2]
    if len(v) > 3:
        env['GF_K4_ABL'] = v[3]
    if len(v) > 2 and v[2] == 'msum':
        env.pop('GF_K4_FORM'); env['GF_K4_MSUM'] = '1'
    if len(v) > 2 and v[2] == 'nopre':
        env.pop('GF_K4_FORM'); env['GF_K4_PRE'] = '0'
    if len(v) > 2 and v[2] == 'gather':
        env.pop('GF_K4_FORM'); env['GF_K4_GATHER'] = '1'
    r = subprocess.run([sys.executable, '-c', CHILD, K], env=env, capture_output=True, text=True)
    print(f'QB={v[0]} WV={v[1]}' + (f' FORM={v[2]}' if len(v) > 2 else '') + (f' ABL={v[3]}' if len(v) > 3 else '') + ': ' + (r.stdout.strip() or r.stderr.strip()[-400:]), flush=True)
