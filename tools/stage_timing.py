"""Per-stage wave cycles of the render-pass kernel (experiment build with -DPGRF_STAGE_TIMING).

    tools/build_variant.sh st "-DPGRF_STAGE_TIMING" 2 2 0 0
    PANOGRF_HIP_LIB=$PWD/panogrf_amd/libpanogrf_hip_st.so PGRF_ALLOW_ABLATION_LIB=1 python tools/stage_timing.py [--precision f16x3]
"""
import argparse, ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from panogrf_amd import engine as E, dist as D, _lib

NAMES = ['ray generation', 'stage A front end (per view)', 'stage B stats + base_fc shared', 'stage C base/vis (per view)',
         'stage D geometry_fc..qkv', 'barrier wait', 'attention', 'fc+LN+density head', 'stage E rgb blend', 'composite: reduce + pixel store',
         'composite: alpha + barrier', 'composite: transmittance scan', 'composite: per-point stores', '', '', '',
         'A: projection + tap setup', 'A: dist decoder MLPs', 'A: compute_prob', 'A: prob_embed', 'A: neuray_fc']
NAMES[1] = 'A: ray_dir_fc + img gather (rest of A)'

ap = argparse.ArgumentParser()
ap.add_argument('--precision', default='f32')
ap.add_argument('--views', type=int, default=2)
a = ap.parse_args()
dev = torch.device('cuda', 0)
H, W, dn = 512, 1024, 64
case = B.synthetic_case(H, W, a.views, dn, seed=0)
case['cfg']['hip_precision'] = a.precision
fr_in = {k: v.to(dev) for k, v in case['frame'].items()}
rr = E.RayRenderer(case['cfg'], case['weights'], device=dev)
coords = D.tile_coords(0, 1, H, W, device=dev)
fr = E.Frame(fr_in['imgs'], fr_in['img_feats'], fr_in['ray_feats'], fr_in['w2c'], fr_in['ref_depth_range'],
             case['frame']['que_c2w'], case['frame']['que_depth_range'], device=dev)
lib = ctypes.CDLL(_lib.library_path())
fn = lib.pgrf_exp_stage_cycles
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
buf = (ctypes.c_ulonglong * 32)()
rr.render_rays(fr, coords); torch.cuda.synchronize()
fn(buf)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); rr.render_rays(fr, coords); e1.record(); torch.cuda.synchronize()
assert fn(buf) == 0
waves = H * W * dn / 32 / 64     # one workgroup in 64 records
tot = sum(buf)
print(f'{a.precision} V={a.views}: {e0.elapsed_time(e1):.2f} ms/frame (instrumented); s_memtime ticks per wave, summed {tot / waves:.0f}')
for i, n in enumerate(NAMES):
    if not n: continue
    print(f'  {n:34s} {buf[i] / waves:9.0f}  {100.0 * buf[i] / tot:5.1f}%')
