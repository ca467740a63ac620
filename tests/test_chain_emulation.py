"""CPU check of the fused pixel-wise chain kernels' index algebra (skyrim_amd/csrc/sfno_chain.hip): the fragment-order weight layouts,
the MFMA 16x16x32 lane mapping, the hidden-chunk -> k-slot correspondence and the perm8 chaining of two expand / contract pairs are
emulated lane by lane in numpy (tools/emulate_chain.py) and must reproduce plain matrix products.  The arithmetic itself (fp16 hi/lo
MFMA terms) is covered by the GPU parity tests."""
import importlib.util
from pathlib import Path

import numpy as np

spec = importlib.util.spec_from_file_location("emulate_chain", Path(__file__).resolve().parent.parent / "tools" / "emulate_chain.py")
emu = importlib.util.module_from_spec(spec)
spec.loader.exec_module(emu)


def test_tail_chain_lane_emulation_matches_matrices(capsys):
    emu.main()
    assert "ok" in capsys.readouterr().out


def test_perm8_is_a_permutation_of_every_group_of_32():
    rho = np.arange(96)
    col = emu.perm8_col(rho)
    assert sorted(col.tolist()) == list(range(96))
    # a lane quad's two fragments (rows 4g..4g+3 of fragment 2bp and of 2bp+1) are 8 consecutive columns 32 bp + 8 g + [0..7]
    for bp in range(3):
        for g in range(4):
            rows = [32 * bp + 4 * g + r for r in range(4)] + [32 * bp + 16 + 4 * g + r for r in range(4)]
            assert emu.perm8_col(np.array(rows)).tolist() == [32 * bp + 8 * g + i for i in range(8)]


def test_prepared_weight_layouts_cover_every_element_once():
    rng = np.random.default_rng(1)
    w1, w2 = rng.normal(size=(96, 64)), rng.normal(size=(64, 96))
    a, b = emu.prep_w1(w1), emu.prep_w2(w2)
    assert a.size == w1.size and b.size == w2.size
    assert np.allclose(np.sort(a.ravel()), np.sort(w1.ravel())) and np.allclose(np.sort(b.ravel()), np.sort(w2.ravel()))
