"""CPU checks of oracle/weights_oracle.py (the restatement of the absent tinycudann / frnn steps; parity unpinned) and of
the host-side half of the hash-grid ABI (no compute calls without a GPU)."""
import ctypes as C
import math

import numpy as np
import torch

from oracle import weights_oracle as wo


def test_grid_meta_matches_library_layout():
    from gaussianprediction_amd import _lib
    from gaussianprediction_amd.weights_ops import HashGridConfigC
    for L, T, N0 in ((16, 19, 16), (16, 12, 16), (8, 15, 4), (1, 10, 16)):
        b = math.exp(math.log(2048 / 16) / 15)
        meta = wo.grid_meta(L, 4, T, N0, b)
        cfg = HashGridConfigC(L, 4, T, N0, b)
        assert int(_lib.lib().gp_hashgrid_table_entries(C.byref(cfg))) == meta["total"]
    bad = HashGridConfigC(16, 2, 19, 16, 1.38)          # only 4 features per level are implemented: loud failure
    assert int(_lib.lib().gp_hashgrid_table_entries(C.byref(bad))) == -1
    assert b"n_features_per_level" in _lib.lib().gp_last_error()


def test_reference_configuration_sizes():
    """[REF scene/gaussian_model.py:370-392]: L=16, F=4, T=2^19, N_min=16, b=exp(ln(2048/16)/15)."""
    meta = wo.grid_meta()
    assert meta["resolutions"][0] == 16 and meta["resolutions"][-1] == 2048
    assert meta["sizes"][0] == 4096 and max(meta["sizes"]) == 1 << 19
    assert all(s % 8 == 0 for s in meta["sizes"])


def test_encoding_interpolates_vertices_and_is_continuous():
    meta = wo.grid_meta(4, 4, 14, 16)
    g = torch.Generator().manual_seed(0)
    table = torch.randn(meta["total"], 4, generator=g, dtype=torch.float64)
    # a point exactly on a level-0 vertex: pos = x*15 + 0.5 integer -> weights (1,0): the vertex's own features
    v = torch.tensor([[2, 3, 5]], dtype=torch.float64)
    x = (v - 0.5) / meta["scales"][0]
    enc = wo.hash_encode(x, table, meta)
    idx = int(v[0, 0] + v[0, 1] * 16 + v[0, 2] * 256) % meta["sizes"][0]
    assert torch.allclose(enc[0, :4], table[idx], atol=1e-6)
    # continuity across a cell boundary
    e = 1e-4
    a = wo.hash_encode(x - e, table, meta)
    b = wo.hash_encode(x + e, table, meta)
    # |d enc / d x| <= scale * (max table difference): a discontinuity would be O(|table|) ~ 1
    assert (a - b).abs().max() < 2 * e * max(meta["scales"]) * 2 * float(table.abs().max())


def test_knn_oracle_against_cdist():
    rng = np.random.default_rng(0)
    p = rng.normal(size=(500, 35)).astype(np.float32)
    q = rng.normal(size=(120, 35)).astype(np.float32)
    idx, d2 = wo.knn(p, q, 6)
    ref = torch.cdist(torch.tensor(p, dtype=torch.float64), torch.tensor(q, dtype=torch.float64)).pow(2)
    ri = torch.topk(ref, 6, dim=1, largest=False).indices.numpy()
    assert (idx == ri).mean() > 0.999                      # float32 vs float64 may swap genuine near-ties
    assert np.all(np.diff(d2, axis=1) >= 0)
