"""Device-side weight packer (pgrf_pack_weights_device, csrc/pack_device.hip) against the host packer, bit for bit.

The reference has no packing step: it evaluates its live nn.Modules (network/renderer.py:223-304), so after every optimizer step the
next forward sees the new weights.  Here the fused kernel reads an MFMA-ordered blob that must be rebuilt from the changed
parameters; in training that rebuild runs on the GPU."""
import ctypes as C

import pytest
import torch

from panogrf_amd import _lib
from panogrf_amd import engine as E


@pytest.mark.parametrize('use_vis', [0, 1])
@pytest.mark.parametrize('max_dn', [128, 48])
def test_source_map_reproduces_the_host_packer(use_vis, max_dn):
    # host evaluation of the map the device kernel walks (no GPU involved)
    assert _lib.lib.pgrf_pack_map_selftest(use_vis, max_dn) == 0, _lib.lib.pgrf_last_error()


def _weights(seed):
    """every raw tensor of one pass (vis_decoder included), shapes from the reference's layer table"""
    g = torch.Generator().manual_seed(seed)
    shapes = {'0.weight': None}
    sd = {}
    dims = {'dist_decoder.mean_decoder': [(32, 32), (32, 32), (2, 32)], 'dist_decoder.var_decoder': [(32, 32), (32, 32), (2, 32)],
            'dist_decoder.aw_decoder': [(32, 32), (32, 32), (1, 32)], 'dist_decoder.vis_decoder': [(32, 32), (32, 32), (1, 32)]}
    named = {n: None for n in E.raw_names()}
    for n in named:
        stem, idx, kind = n.rsplit('.', 2)
        if stem in dims:
            o, i = dims[stem][int(idx) // 2]
            shape = (o, i) if kind == 'weight' else (o,)
        else:
            shape = None
        named[n] = shape
    # the aggregation net's shapes: from the synthetic case (it has no vis decoder, hence the table above)
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import synthetic_case
    case = synthetic_case(32, 64, 2, 8, seed=seed, fine=False)
    for n, shape in named.items():
        if shape is None:
            shape = tuple(case['weights'][n].shape)
        sd[n] = 0.3 * torch.randn(shape, generator=g)
    return sd


@pytest.mark.gpu
@pytest.mark.parametrize('use_vis', [False, True])
def test_device_pack_equals_host_pack(use_vis):
    sd = _weights(3)
    host = E.pack_pass_weights(sd, 'dist_decoder', 'agg_net', use_vis)
    assert not host.is_cuda
    dev_sd = {k: v.cuda() for k, v in sd.items()}
    dev = E.pack_pass_weights(dev_sd, 'dist_decoder', 'agg_net', use_vis)
    assert dev.is_cuda
    a, b = host.view(torch.int32), dev.cpu().view(torch.int32)
    bad = (a != b).nonzero()
    assert bad.numel() == 0, f'{bad.numel()} of {a.numel()} blob words differ, first at {int(bad[0])}'
    # a second call (cached map) after the weights moved
    for v in dev_sd.values():
        v.mul_(1.25)
    host2 = E.pack_pass_weights({k: v.cpu() for k, v in dev_sd.items()}, 'dist_decoder', 'agg_net', use_vis)
    dev2 = E.pack_pass_weights(dev_sd, 'dist_decoder', 'agg_net', use_vis)
    assert torch.equal(host2.view(torch.int32), dev2.cpu().view(torch.int32))


@pytest.mark.gpu
def test_device_pack_rejects_missing_tensor():
    n = _lib.lib.pgrf_raw_count()
    arr = (C.c_void_p * n)()
    blob = torch.empty(_lib.lib.pgrf_blob_floats(), device='cuda')
    rc = _lib.lib.pgrf_pack_weights_device(arr, n, 1, 128, C.c_void_p(blob.data_ptr()), None)
    assert rc != 0 and b'null' in _lib.lib.pgrf_last_error()
