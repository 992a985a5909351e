"""SiamFC probe (§8f rank 4): heads.  Oracle pinned against responses of the reference classes
(tests/golden/siamfc_heads.npz); the HIP heads (vfs_xcorr_fwd + 1x1 vfs_conv_fwd) against the oracle on the same
bf16-rounded operands.  backend=emu (CPU) / gpu."""
import os

import numpy as np
import pytest
import torch

from oracle import siamfc_oracle as SO
from oracle import vfs_oracle as O

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'siamfc_heads.npz'))
TAGS = sorted({k.split('/')[0] for k in G.files})


def _inputs(tag):
    nz, nx, c, hz, h = (int(v) for v in G[tag + '/shape'])
    return nz, nx, c, O.fill_tensor([nz, c, hz, hz], 3, scale=1.5), O.fill_tensor([nx, c, h, h + 2], 4, scale=1.5)


@pytest.mark.parametrize('tag', TAGS)
def test_oracle_heads_match_reference(tag):
    nz, nx, c, z, x = _inputs(tag)
    with torch.no_grad():
        got = SO.SiamFC(out_scale=0.001)(z, x).numpy()
        head = SO.SiamConvFC(c, 2 * c, out_scale=0.01)
        assert list(head.state_dict().keys()) == [str(k) for k in G[tag + '/keys']]
        O.fill_state_dict_(head, seed=21)
        got2 = head(z, x).numpy()
    assert np.allclose(got, G[tag + '/siamfc'], rtol=1e-5, atol=1e-7)
    assert np.allclose(got2, G[tag + '/siamconvfc'], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('tag', TAGS)
def test_hip_heads_match_oracle(backend, tag):
    import vfs_amd
    nz, nx, c, z, x = _inputs(tag)
    zr, xr = O.round_bf16(z), O.round_bf16(x)
    dev = backend.dev
    with torch.no_grad():
        want = SO.SiamFC(out_scale=0.001)(zr, xr)
        got = vfs_amd.SiamFC(out_scale=0.001)(zr.to(dev), xr.to(dev)).cpu()
        assert got.shape == want.shape
        assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max()) + 1e-9      # fp32 sums of identical bf16 products
        ref = SO.SiamConvFC(c, 2 * c, out_scale=0.01)
        O.fill_state_dict_(ref, seed=21)
        head = vfs_amd.SiamConvFC(c, 2 * c, out_scale=0.01)
        head.load_state_dict(ref.state_dict())
        head.to(dev)
        got2 = head(zr.to(dev), xr.to(dev)).cpu()
        want2 = ref(zr, xr)
        # bf16 weights and bf16 conv outputs in the HIP head: one rounding of each 1x1 conv
        assert float((got2 - want2).norm() / want2.norm()) < 1.5e-2
    with pytest.raises(NotImplementedError):
        vfs_amd.SiamConvFC(c, c, kernel_size=3)
