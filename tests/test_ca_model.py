"""The numpy model of the Gram-based panel (tests/ca_model.py; device code: csrc/stmmqr_capanel.hip) against the restated
reference front on the real fronts of a fixture: the algorithm -- one Gram matrix per panel, downdated per column,
refreshed when a column has lost 1/K of its norm -- must reproduce the integer outputs (rank, staircase, dead columns)
and an orthogonal factorization to rounding."""
import pytest

import ca_model


@pytest.mark.parametrize("name,mincols", [("syn_grid3d", 16), ("lns_3937", 48), ("bcsstk14", 64)])        # (lns_3937: rank 1822 of 3908)
def test_gram_panel_model_matches_reference_front(name, mincols):
    res, stats = ca_model.main(name, 32.0, mincols)
    assert res and stats["panels"] > 0
    for fm, fn, fp, same_int, dn, low, rerr, nrefresh, npanels in res:
        assert same_int, (fm, fn, fp)
        assert low <= 1e-13 and rerr <= 1e-13, (fm, fn, fp, low, rerr)
