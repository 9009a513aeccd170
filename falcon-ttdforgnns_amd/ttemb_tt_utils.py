"""The two core initialisers of the reference's `tt_utils` module, on the MI355X.  Named `ttemb_tt_utils` so that this
package directory can sit in front of the reference tree on sys.path without shadowing the reference's own `tt_utils`
(every driver does `from tt_utils import *` for its argument parser): a driver that wants these versions imports
`from ttemb_tt_utils import get_ortho, tt_matrix_decomp`.

Only what feeds the TT layer is provided: `get_ortho` (tt_utils.py:117-157) and `tt_matrix_decomp`
(tt_utils.py:159-201), with the reference's argument order and return types, so that
`gnn_model.py:127-181` (`--init ortho|dortho|eigen`) keeps working.  The argument parser, eigen solver
and logging helpers of that module belong to the drivers and are out of scope.
"""
from __future__ import annotations

import numpy as np
import torch

import ttemb_init

_DEVICE = "cuda" if torch.cuda.is_available() else "cpu"


def get_ortho(tt_ranks, tt_p_shapes, tt_q_shapes):
    """-> list of numpy float32 [1, p_t, R_t q_t R_{t+1}] (callers wrap them in torch.tensor)."""
    cores = ttemb_init.ortho_cores(tt_ranks, tt_p_shapes, tt_q_shapes, device=_DEVICE)
    return [c.cpu().numpy() for c in cores]


def tt_matrix_decomp(matrix, tt_ranks, tt_p_shapes, tt_q_shapes):
    """-> (list of torch float32 cores [1, p_t, -1] on the CPU, ranks) like the reference."""
    x = torch.as_tensor(np.asarray(matrix, dtype=np.float32) if not torch.is_tensor(matrix) else matrix)
    cores, ranks = ttemb_init.tt_svd_cores(x.to(_DEVICE), tt_ranks, list(tt_p_shapes), list(tt_q_shapes))
    return [c.cpu() for c in cores], ranks
