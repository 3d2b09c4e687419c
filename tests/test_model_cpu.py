"""Host-side behaviour of the model mirror (model/NeRF.py of the reference: NeRF.py:55-78) that needs no GPU."""
import copy
import io
import pickle

import pytest
import torch

from nerf_pytorch_paeng_amd._lib import MiNerfError
from nerf_pytorch_paeng_amd.model import NeRF
from nerf_pytorch_paeng_amd.model.NeRF import NeRFModule


def _parents(m):
    return m.model_coarse._parent(), m.model_fine._parent()


def test_submodule_back_references_follow_copies():
    """``model.model_coarse(x)`` routes through the parent NeRF (a weak back reference).  A deepcopy (an EMA copy), a pickle
    round trip and ``torch.save(model)`` must bind the COPY's sub-modules to the copy, not to the original."""
    m = NeRF(4, 128, 63, 27)
    assert _parents(m) == (m, m)
    c = copy.deepcopy(m)
    assert _parents(c) == (c, c) and _parents(m) == (m, m)
    del m                                              # the copy must not depend on the original staying alive
    assert _parents(c) == (c, c)
    p = pickle.loads(pickle.dumps(c))
    assert _parents(p) == (p, p)
    buf = io.BytesIO()
    torch.save(c, buf)
    buf.seek(0)
    r = torch.load(buf, weights_only=False)
    assert _parents(r) == (r, r)
    for k, v in c.state_dict().items():
        assert torch.equal(v, r.state_dict()[k]) and torch.equal(v, p.state_dict()[k])
    assert "_parent" not in c.model_fine.__getstate__()


def test_orphan_submodule_refuses():
    with pytest.raises(MiNerfError):
        NeRFModule(4, 128, 63, 27)(torch.zeros(2, 90))
