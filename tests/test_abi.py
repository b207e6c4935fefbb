"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol the
header declares; the Python surface keeps the reference's names and state_dict keys."""
import os
import re
import subprocess

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "ustrun.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ustrun_[A-Za-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from ustrun import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import sys
        sys.path.insert(0, ROOT)
        import __graft_entry__ as g
        g.build()
    syms = header_symbols()
    assert len(syms) >= 30
    assert sorted(_lib.SIGNATURES) == syms          # binding table == header
    h = _lib.lib()
    for s in syms:
        assert hasattr(h, s), s
    assert h.ustrun_version() == 100
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (ustrun_\w+)", out))
    assert set(syms) <= exported


def test_error_path_without_gpu_work():
    from ustrun import _lib
    h = _lib.lib()
    rc = h.ustrun_pack_conv3x3(None, 0, 0, None, None, 0, None)     # argument check fails before any launch
    assert rc != 0 and b"pack_conv3x3" in h.ustrun_last_error()


def test_state_dict_keys_match_reference_surface():
    from networks.unet_model import UNet
    from oracle import unet_ref as U
    torch.manual_seed(3)
    m = UNet(n_channels=3, n_classes=2)
    torch.manual_seed(3)
    sd = U.make_state_dict(3, 2)
    msd = m.state_dict()
    assert list(msd.keys()) == list(sd.keys()) and len(msd) == 118
    for k in sd:
        assert msd[k].shape == sd[k].shape, k
        assert torch.equal(msd[k], sd[k]), k          # same RNG consumption order as the reference
    assert [n for n, _ in m.named_parameters()] == U.param_keys(sd)
    assert (m.n_channels, m.n_classes, m.bilinear) == (3, 2, False)
    assert sum(p.numel() for p in m.parameters()) == 31037698
