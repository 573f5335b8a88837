"""Regenerate the measured-numbers blocks of DESIGN.md and README.md from the files under profiles/ (tools/install_profiles.sh):
everything between `<!-- BEGIN GENERATED:measured -->` and `<!-- END GENERATED:measured -->` is replaced, so the documents
quote the committed profiles, never prose typed earlier in a round.   python tools/summarize_profiles.py r03"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else 'r03'
P = lambda name: os.path.join(ROOT, 'profiles', f'{TAG}_{name}')


def load_json(name):
    try:
        txt = open(P(name)).read()
        line = [ln for ln in txt.splitlines() if ln.startswith('{')][-1]
        return json.loads(line)
    except Exception:
        return None


def read(name):
    try:
        return open(P(name)).read().strip()
    except Exception:
        return None


def pmc(name):
    vals = {}
    for ln in (read(name) or '').splitlines():
        t = ln.split()
        if len(t) >= 3 and not ln.startswith('#'):
            try:
                vals[t[1]] = float(t[2])
            except ValueError:
                pass
    return vals


def kernel_avg_ms(name, needle):
    try:
        for row in csv.DictReader(open(P(name))):
            if needle in row['Name']:
                return float(row['AverageNs']) / 1e6, int(row['Calls'])
    except Exception:
        pass
    return None, None


out = []
b = load_json('bench.json')
if b:
    r = b['roofline']
    out.append(f'**Headline (`profiles/{TAG}_bench.json`, the plain `python bench.py` line):** {b["ms_per_step"]:.2f} ms per 512x1024 frame = '
               f'{b["value"] / 1e6:.2f} M rays/s; `render_pass_kernel` {r["kernel_ms"]:.2f} ms per launch (HIP events) = {r["achieved"]:.1f} TFLOP/s '
               f'algorithmic = **{r["frac"] * 100:.1f} %** of the {r["peak"]} TFLOP/s f32 MFMA peak.')
    k_ms, calls = kernel_avg_ms('bench_kernel_stats.csv', 'render_pass_kernel<2, 2')
    if k_ms is None:
        k_ms, calls = kernel_avg_ms('bench_kernel_stats.csv', 'render_pass_kernel')
    if k_ms:
        out.append(f'rocprofv3 `--kernel-trace --stats` of the same command (`profiles/{TAG}_bench_kernel_stats.csv`): {k_ms:.2f} ms average over {calls} launches.')
    c = pmc('render_pass_pmc.txt')
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
        traffic = (2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024
        hit = c.get('TCC_HIT_sum', 0) / max(c.get('TCC_HIT_sum', 0) + c.get('TCC_MISS_sum', 0), 1)
        out.append(f'HBM-side traffic per launch (`profiles/{TAG}_render_pass_pmc.txt`, separate `--pmc` passes; FETCH_SIZE doubled per the gfx950 note of '
                   f'MI355X_MICROARCH.md): 2 x {c["FETCH_SIZE"] * 1024 / 1e9:.2f} + {c["WRITE_SIZE"] * 1024 / 1e9:.2f} = **{traffic / 1e9:.2f} GB** '
                   f'({traffic / (r["kernel_ms"] * 1e-3) / 1e12:.2f} TB/s); L2 hit rate TCC_HIT / (HIT + MISS) = {hit * 100:.1f} %; '
                   f'MFMA-pipe busy SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8: the counter sums the 8 XCDs) = '
                   f'{c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * max(c.get("GRBM_GUI_ACTIVE", 1), 1) / 8.0) * 100:.0f} %.')
    if str(r.get('traffic_source', '')).startswith('measured by this run') and r.get('traffic'):
        out.append(f'The line\'s own `roofline.traffic` (two child passes of the run under `rocprofv3 --pmc`, bench.py `live_hbm_traffic`): {r["traffic"] / 1e9:.2f} GB per launch.')
    e = b.get('torch_eager_gpu_baseline')
    if e:
        out.append(f'Eager PyTorch-ROCm port of the reference path on the same GPU (driver-run, bounded strip: {e.get("sample", "")}): '
                   f'{e.get("value_ray_batch_2048", 0) / 1e3:.1f} k rays/s at 2 048 rays per batch, {e.get("value_ray_batch_4096", 0) / 1e3:.1f} k at 4 096 -> '
                   f'**{e["speedup_hip_over_eager"]:.0f} x**; PSNR(HIP, eager) = {e.get("psnr_hip_vs_eager_db", float("nan")):.1f} dB.')
    cb = b.get('cpu_baseline')
    if cb:
        out.append(f'CPU oracle: {cb["value"] / 1e3:.2f} k rays/s on {cb["cores"]} threads ({cb["sample"]}).')
    ps = b.get('psnr')
    if ps:
        out.append(f'Parity on the oracle\'s rays: float PSNR {ps["float_vs_oracle_db"]:.1f} dB, WS-PSNR {ps["ws_vs_oracle_db"]:.1f} dB, uint8 PSNR '
                   f'{ps["uint8_vs_oracle_db"] if ps["uint8_vs_oracle_db"] is None else round(ps["uint8_vs_oracle_db"], 1)} dB, '
                   f'|PSNR(HIP, GT) - PSNR(oracle, GT)| = {ps["abs_delta_vs_gt_db"]:.1e} dB ({ps["rays"]} rays).')
    ee = b.get('end_to_end')
    if ee:
        out.append(f'Whole `renderer(data)` frames for the bench configuration: {ee["ms_per_frame_everything_per_frame"]:.1f} ms with the MVS stage and the '
                   f'encoders recomputed per frame, {ee["ms_per_frame_source_maps_reused"]:.1f} ms on a trajectory against fixed source views.')
    out.append('')

bt = load_json('bench_train.json')
if bt:
    r = bt['roofline']
    eg = bt.get('torch_eager_gpu_baseline') or {}
    out.append(f'**Training step (`profiles/{TAG}_bench_train.json`, `python bench.py --train`):** {bt["config"]["rays_per_step_per_rank"]} rays x (64 + 64) samples, '
               f'forward + backward + Adam of the per-ray networks: {bt["ms_per_step"]:.2f} ms per step = {bt["value"] / 1e3:.1f} k rays/s = {r["achieved"]:.1f} TFLOP/s '
               f'algorithmic (3 x the forward) = **{r["frac"] * 100:.1f} %** of the f32 MFMA peak'
               + (f'; eager-autograd port of the reference graph on the same GPU {eg["ms_per_step"]:.0f} ms per step -> {eg["speedup_hip_over_eager"]:.1f} x.' if 'ms_per_step' in eg else '.'))
    rh = bt.get('roofline_hbm') or {}
    if 'frac' in rh:
        out.append(f'The step is bound by the HBM side of its materialised activations: {rh["traffic"] / 1e9:.2f} GB per step (all kernels; two child passes of the run '
                   f'under `rocprofv3 --pmc`) = {rh["achieved"] / 1e3:.2f} TB/s = **{rh["frac"] * 100:.0f} %** of 8 TB/s (`roofline_hbm` on the same line).')
    out.append('')

rows = []
for name, label in [('bench_config3.json', 'config 3: + D=64 cost volume 128x256 + UNet2 per source view (`--with-cost-volume`)'),
                    ('bench_config3_costregnet.json', 'config 3 with CostRegNet (`--reg costregnet`)'),
                    ('bench_config4_fine.json', 'config 4 on one GPU: 64 coarse + 64 fine (`--fine`)'),
                    ('bench_views3.json', '3 source views (`--views 3`)'), ('bench_views4.json', '4 source views (`--views 4`)'),
                    ('bench_config5_1gpu.json', 'config 5 on one GPU: 1024x2048, 4 views'),
                    ('bench_f16x3.json', 'opt-in split-f16 products (`--precision f16x3`, own dtype)')]:
    d = load_json(name)
    if not d:
        continue
    extra = ''
    if 'cost_volume_stage' in d:
        cs = d['cost_volume_stage']
        extra = f'; cost-volume stage {cs["ms_per_step"]:.2f} ms'
        if 'roofline' in cs:
            extra += (f' (per view: build {cs["roofline_build"]["kernel_ms"] * 1e3:.0f} us = {cs["roofline_build"]["frac"] * 100:.0f} % of 8 TB/s, '
                      f'{cs["regulariser"]} {cs["roofline"]["kernel_ms"]:.2f} ms = {cs["roofline"]["frac"] * 100:.0f} % of its MFMA roof)')
    if 'psnr' in d:
        extra += f'; PSNR vs oracle {d["psnr"]["float_vs_oracle_db"]:.1f} dB, |dPSNR vs GT| {d["psnr"]["abs_delta_vs_gt_db"]:.1e} dB'
    rows.append(f'| {label} | {d["ms_per_step"]:.2f} | {d["value"] / 1e6:.2f} | {d["roofline"]["frac"] * 100:.1f} %{extra} | `{TAG}_{name}` |')
if rows:
    out.append('| configuration (one MI355X) | ms / step | M rays/s | render kernel vs its roof | file |')
    out.append('|---|---|---|---|---|')
    out += rows
    out.append('')
for name, title in [('costvolume_timings.txt', 'Cost-volume side (`tools/time_cost_volume.py`)'),
                    ('cost_volume_in_stage.txt', 'Cost-volume build inside a polluted-cache stage loop (`tools/cv_stage_nt.py`: per step 1 GB read-modify-write, hypotheses, BUILD, regulariser)'),
                    ('costregnet_layer_trace_128x256.txt', 'CostRegNet per-layer kernel durations at 128x256x64'),
                    ('costreg_conv_pmc.txt', "SQ counters of CostRegNet's conv kernels (`tools/pmc_conv.sh`)"),
                    ('winograd_vs_direct_layers.txt', 'Winograd kernels per UNet2 layer shape (`tools/check_wino.py time`: second generation | first generation | direct kernel)'),
                    ('unet2_layer_trace_128x256.txt', 'Per-launch durations of one UNet2 forward at 128x256x64 (`tools/layer_trace.sh`)'),
                    ('frame_timing.txt', 'Whole frames, shipped configuration (`tools/time_frame.py`)'),
                    ('whole_frame_rank_shares.txt', 'Per-rank shares of one frame under the 8-GPU split (`tools/time_frame_shares.py`, one GPU)'),
                    ('train_demo.txt', 'Training steps through the HIP forward / backward (`tools/train_demo.py`)'),
                    ('views4_mem_pmc.txt', 'HBM-side counters of the 4-view kernel (`tools/pmc_mem.sh`)')]:
    t = read(name)
    if t:
        out.append(f'{title} -- `profiles/{TAG}_{name}`:')
        out.append('```')
        out += [ln[:260] for ln in t.splitlines() if not ln.startswith('W2026') and 'rocprofv3' not in ln][:26]
        out.append('```')
block = '\n'.join(out).rstrip() + '\n'

BEGIN, END = '<!-- BEGIN GENERATED:measured -->', '<!-- END GENERATED:measured -->'
for doc in ('DESIGN.md', 'README.md'):
    path = os.path.join(ROOT, doc)
    s = open(path).read()
    if BEGIN not in s:
        print(f'{doc}: no generated block (add {BEGIN} ... {END})')
        continue
    head, rest = s.split(BEGIN, 1)
    _, tail = rest.split(END, 1)
    body = block if doc == 'DESIGN.md' else '\n'.join(block.splitlines()[:12]) + '\n'
    open(path, 'w').write(head + BEGIN + f'\n(generated by tools/summarize_profiles.py from profiles/{TAG}_*; do not edit by hand)\n\n' + body + END + tail)
    print(f'{doc}: block regenerated ({len(body.splitlines())} lines)')
