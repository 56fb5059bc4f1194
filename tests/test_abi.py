"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/geoformer_hip.h declares, the ctypes table agrees with the header, argument validation
reports errors through status codes, and the product refuses to run without a GPU."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'geoformer_hip.h')


def header_functions():
    txt = open(HEADER).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    out = {}
    for m in re.finditer(r'^\s*(?:const\s+)?(?:int|size_t|void|char\s*\*|const char\*)\s*\*?\s*(gf_\w+)\s*\(([^;]*?)\)\s*;', txt, flags=re.M | re.S):
        args = [a.strip() for a in m.group(2).replace('\n', ' ').split(',')]
        out[m.group(1)] = 0 if args == ['void'] else len(args)
    return out


@pytest.fixture(scope='module')
def lib():
    from geoformer_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib


def test_every_declared_symbol_is_exported_and_bound(lib):
    decl = header_functions()
    assert len(decl) >= 20
    h = lib.lib()
    for name, nargs in decl.items():
        assert hasattr(h, name), f'{name} declared in the header but not exported'
        assert name in lib.SIGNATURES, f'{name} has no ctypes signature'
        assert len(lib.SIGNATURES[name][1]) == nargs, f'{name}: header has {nargs} arguments'
    assert set(lib.SIGNATURES) == set(decl), set(lib.SIGNATURES) ^ set(decl)
    syms = subprocess.run(['nm', '-D', '--defined-only', lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r' T (gf_\w+)', syms))
    assert set(decl) <= exported


def test_abi_version_and_workspace_queries(lib):
    h = lib.lib()
    # the binding checks the version at load; the header's macro, the library and the binding agree
    import re as _re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'geoformer_hip.h')).read()
    assert h.gf_abi_version() == lib.ABI_VERSION == int(_re.search(r'#define GF_ABI_VERSION (\d+)', hdr).group(1))
    assert h.gf_dual_softmax_workspace_bytes(1, 6400, 6400) > 6400 * 8
    assert h.gf_dual_softmax_workspace_bytes(0, 1, 1) == 0
    assert h.gf_linear_attention_workspace_bytes(2, 6400, 8, 32) > 0
    assert h.gf_self_attention_workspace_bytes(1, 6400, 1) == 2 * 6400 * 256 * 2
    assert h.gf_ransac_workspace_bytes(2, 1024) >= 2 * 1024 * (72 + 4)
    assert h.gf_fine_match_workspace_bytes(3000) > 3000 * 4


def test_argument_errors_come_back_as_status_codes(lib):
    h = lib.lib()
    # null pointers / bad dtype are rejected before anything touches a device
    rc = h.gf_dual_softmax_match(None, None, 0, 1, 8, 8, 64, None, None, 0.1, 0.2, 0, 1, 1, 8.0, None, None, None, None, None,
                                 None, None, None, None, None, None, 0, None)
    assert rc == -1 and b'null pointer' in h.gf_last_error()
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    rc = h.gf_linear(p, 64, 64, None, 0, 0, p, None, None, 0, 9, None, None, 1e-5, None, 0, None, 0, p, 64, 1, 1, 64, None)
    assert rc == -1 and b'epilogue' in h.gf_last_error()
    rc = h.gf_linear(p, 64, 48, None, 0, 0, p, None, None, 0, 0, None, None, 1e-5, None, 0, None, 0, p, 64, 1, 1, 64, None)
    assert rc == -1 and b'multiples' in h.gf_last_error()
    rc = h.gf_window_cross_attention(p, p, p, 1, 1, 8, 8, 8, 32, 256, 256, 256, p, 25, None, p, None)
    assert rc == -1 and b'nhead=4' in h.gf_last_error()


def test_no_cpu_fallback():
    from geoformer_amd import ops, _lib
    from geoformer_amd.model.full_model import GeoFormer
    from geoformer_amd.model.cvpr_ds_config import get_default_cfg
    from geoformer_amd.model.geo_config import get_cfg_model
    with pytest.raises(_lib.GeoFormerHipError):
        ops.dual_softmax_match(torch.zeros(1, 8, 64), torch.zeros(1, 8, 64), 0.1, 0.2, (2, 4), (2, 4), 8.0)
    m = GeoFormer(get_default_cfg(), get_cfg_model()).eval()
    with pytest.raises(RuntimeError, match='no CPU path'):
        m({'image0': torch.zeros(1, 1, 64, 64), 'image1': torch.zeros(1, 1, 64, 64)})


def test_config_and_state_dict_contract():
    """geo_config / cvpr_ds_config keys and the 253-tensor state dict of the reference."""
    import geoformer_oracle as O
    from geoformer_amd.model import geo_config, cvpr_ds_config
    from geoformer_amd.model.full_model import GeoFormer
    assert geo_config.default_cfg == O.default_geo_config()
    assert cvpr_ds_config.default_cfg == O.default_loftr_config()
    cfg = cvpr_ds_config.get_default_cfg()
    gc = geo_config.get_cfg_model(); gc['coarse_thr'] = 0.37
    m = GeoFormer(cfg, gc)
    assert cfg['match_coarse']['thr'] == 0.37                       # constructor side effect (full_model.py:31)
    assert m.coarse_matching.border_rm == 0                          # coarse_matching.py:32
    sd = m.state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: tuple(v) for k, v in O.state_dict_schema().items()}
    # load_state_dict strips the Lightning 'matcher.' prefix (full_model.py:125-129) and accepts strict=False
    W = O.make_weights()
    m.load_state_dict({'matcher.' + k: v for k, v in W.items()}, strict=False)
    assert torch.equal(m.state_dict()['loftr_coarse.layers.3.merge.weight'], W['loftr_coarse.layers.3.merge.weight'])
    assert not any(k.endswith('.pe') for k in sd)                    # pe buffers are not persistent
    from geoformer_amd.weights import deterministic_init_
    m2 = deterministic_init_(GeoFormer(cvpr_ds_config.get_default_cfg(), geo_config.get_cfg_model()))
    assert all(torch.equal(m2.state_dict()[k], W[k]) for k in W)


def test_reference_module_paths_resolve_through_the_alias_packages():
    """inference.py:6-9 / eval_tool/immatch/modules/geoformer.py:6-10 import `model.full_model`, `model.geo_config` and
    `model.loftr_src.loftr.utils.cvpr_ds_config`: with geoformer_amd/compat on the path those lines resolve to this package unchanged."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('from model.full_model import GeoFormer\n'
            'from model.geo_config import default_cfg as g\n'
            'from model.loftr_src.loftr.utils.cvpr_ds_config import default_cfg as c\n'
            "assert GeoFormer.__module__ == 'geoformer_amd.model.full_model'\n"
            "assert {'layer_names', 'nhead', 'coarse_thr', 'fine_temperature', 'fine_thr', 'window_size', 'topk'} <= set(g)\n"
            "assert {'backbone_type', 'resolution', 'coarse', 'match_coarse', 'fine'} <= set(c)\n"
            "print('ok')\n")
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.path.join(root, 'geoformer_amd', 'compat'))
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == 'ok', r.stderr[-800:]
