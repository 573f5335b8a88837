"""Gradients of a network pass against golden vectors from the REFERENCE's own autograd graph (tests/golden/make_grad_fixtures.py:
network/renderer.py:223-304 differentiated in the build container).  CPU: the oracle restatement under torch.autograd;
`-m gpu`: pgrf_render_pass_backward.  Bar: relative L2 <= 5e-4 per tensor (tensors whose reference gradient is round-off --
rgb_fc.4.bias: a softmax over the views ignores a common shift -- are compared absolutely)."""
import numpy as np
import pytest
import torch

from tests import fixtures as FX

GRAD_FIXTURES = [('render_grad_v2_dn32', 'render_stages_v2_dn32'), ('render_grad_v4_dn32', 'render_stages_v4_dn32')]
REL_L2 = 5e-4
PASSES = [('coarse', 'dist_decoder', 'agg_net'), ('fine', 'fine_dist_decoder', 'fine_agg_net')]


def _err(got, ref):
    """relative L2, or -- where the reference gradient is pure round-off -- the absolute norm scaled by 1e-4"""
    nr = float(torch.linalg.norm(ref.double()))
    d = float(torch.linalg.norm((got.double() - ref.double())))
    return d / nr if nr > 1e-5 else d / 1e-4 * REL_L2


def _load(gname):
    z = np.load(FX.GOLDEN + '/' + gname + '.npz')
    return {k: torch.from_numpy(z[k]) for k in z.files if k not in ('source',)}


@pytest.mark.parametrize('gname,sname', GRAD_FIXTURES)
def test_oracle_autograd_vs_reference_gradients(gname, sname):
    from oracle import render_oracle as O
    frame, w, cfg, g = FX.load(sname)
    G = _load(gname)
    use_vis = cfg['dist_decoder_cfg']['use_vis']
    for tag, dd, ag in PASSES:
        keys = [k for k in w if k.startswith((dd + '.', ag + '.'))]
        wl = {k: (w[k].clone().requires_grad_(True) if k in keys else w[k]) for k in w}
        fr = dict(frame, img_feats=frame['img_feats'].clone().requires_grad_(True), ray_feats=frame['ray_feats'].clone().requires_grad_(True))
        out = O.render_pass(wl, dd, ag, fr, frame['coords'], g[tag]['depth'], {'use_vis': use_vis})
        up = {k[len(f'up.{tag}.'):]: v for k, v in G.items() if k.startswith(f'up.{tag}.')}
        loss = sum((out[k] * up[k]).sum() for k in up)
        assert abs(float(loss.detach()) - float(G[f"loss.{tag}"])) < 1e-4 * max(1.0, abs(float(G[f'loss.{tag}'])))
        grads = torch.autograd.grad(loss, [wl[k] for k in keys] + [fr['img_feats'], fr['ray_feats']], allow_unused=True)
        worst = ('', 0.0)
        for k, gt in zip(keys + ['img_feats', 'ray_feats'], grads):
            ref = G[f'grad.{tag}.{k}']
            gt = torch.zeros_like(ref) if gt is None else gt
            e = _err(gt, ref)
            worst = max(worst, (k, e), key=lambda t: t[1])
            assert e < REL_L2, (gname, tag, k, e)
        assert len(keys) == 62


@pytest.mark.gpu
@pytest.mark.parametrize('gname,sname', GRAD_FIXTURES)
def test_hip_backward_vs_reference_gradients(gname, sname):
    from panogrf_amd import engine as E
    frame, w, cfg, g = FX.load(sname)
    G = _load(gname)
    use_vis = cfg['dist_decoder_cfg']['use_vis']
    fr = E.Frame(frame['imgs'], frame['img_feats'], frame['ray_feats'], frame['w2c'], frame['ref_depth_range'], frame['que_c2w'],
                 frame['que_depth_range'])
    names = E.raw_names()
    for tag, dd, ag in PASSES:
        up = {k[len(f'up.{tag}.'):]: v for k, v in G.items() if k.startswith(f'up.{tag}.')}
        raw = E.raw_weight_tensors(w, dd, ag, use_vis, fr.device)
        g_raw, g_if, g_rf, fwd = E.render_pass_backward(fr, raw, use_vis, frame['coords'], g[tag]['depth'], up, want_forward=True)
        loss = sum(float((fwd[k].cpu().double() * up[k].double()).sum()) for k in up)
        assert abs(loss - float(G[f'loss.{tag}'])) < 2e-4 * max(1.0, abs(float(G[f'loss.{tag}'])))
        checked = 0
        for n_, gt in zip(names, g_raw):
            key = (dd + n_[len('dist_decoder'):]) if n_.startswith('dist_decoder.') else (ag + n_[len('agg_net'):])
            if gt is None:
                assert '.vis_decoder.' in key and not use_vis
                continue
            e = _err(gt.cpu(), G[f'grad.{tag}.{key}'])
            assert e < REL_L2, (gname, tag, key, e)
            checked += 1
        assert checked == 62
        assert _err(g_if.cpu(), G[f'grad.{tag}.img_feats']) < REL_L2, (gname, tag, 'img_feats')
        assert _err(g_rf.cpu(), G[f'grad.{tag}.ray_feats']) < REL_L2, (gname, tag, 'ray_feats')
