"""Cycle-weighted instruction census of the render-pass kernel per stage, from the assembly of a -DPGRF_STAGE_MARK build:

    python tools/isa_stage_census.py [KT] [V] [extra -D flags]

Costs per wave instruction (cycles of SIMD time, two waves per SIMD beside f32 MFMAs) are the ones tools/ubench/issue_model.hip
measured on MI355X (profiles/r05_ubench_issue_model.txt): the f32 MFMA does not overlap with VALU work of either wave, so a stage's
time is close to the sum over its instructions."""
import collections, os, re, subprocess, sys
KT = sys.argv[1] if len(sys.argv) > 1 else '2'
V = sys.argv[2] if len(sys.argv) > 2 else '2'
extra = sys.argv[3:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = '/tmp/isa_census_%s_%s.s' % (KT, V)
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-Wno-unused-result', '-I' + root + '/include',
                       '-DPGRF_ABLATION_BUILD', '-DPGRF_STAGE_MARK', '-DPGRF_KT=' + KT, '-DPGRF_V=' + V, '-DPGRF_DBG=0', '-DPGRF_PREC=0', *extra,
                       '-S', '--cuda-device-only', root + '/panogrf_amd/csrc/render_pass_inst.hip', '-o', out], stderr=subprocess.DEVNULL)
FAST = 2.45; SLOW = 4.2; TRANS = 7.9
COST = {'v_mul_f32': FAST, 'v_add_f32': FAST, 'v_sub_f32': FAST, 'v_subrev_f32': FAST, 'v_fmaak_f32': 2.6, 'v_fmamk_f32': 2.6,
        'v_exp_f32': TRANS, 'v_rcp_f32': TRANS, 'v_log_f32': TRANS, 'v_rsq_f32': TRANS, 'v_sqrt_f32': TRANS, 'v_sin_f32': TRANS, 'v_cos_f32': TRANS,
        'v_pk_fma_f32': 4.6, 'v_pk_mul_f32': 4.6, 'v_pk_add_f32': 4.6, 'v_permlane32_swap_b32': 8.8, 'v_permlane16_swap_b32': 8.8,
        'v_mfma_f32_32x32x2_f32': 64, 'v_mfma_f32_16x16x4_f32': 32, 'v_mfma_f32_4x4x1_16b_f32': 8,
        'ds_read_b128': 9.5, 'ds_read_b32': 3.4, 'ds_read2_b32': 5, 'ds_write_b32': 4, 'ds_bpermute_b32': 8, 'global_load_dwordx4': 8, 'global_load_dwordx3': 8,
        'global_load_dword': 4, 'global_store_dword': 4, 's_nop': 1.5}
names = {0: 'ray generation', 16: 'A: projection + tap setup', 17: 'A: dist decoder MLPs', 18: 'A: compute_prob', 19: 'A: prob_embed', 20: 'A: neuray_fc',
         1: 'A: ray_dir_fc + img gather', 2: 'B: stats + base_fc shared', 3: 'C: base/vis per view', 4: 'D: geometry_fc..qkv', 5: 'barrier wait',
         6: 'attention', 7: 'fc+LN+density head', 8: 'E: rgb blend', 10: 'composite: alpha+barrier', 11: 'composite: scan', 12: 'composite: stores', 9: 'composite: reduce'}
txt = open(out).read()
sym = [l for l in txt.split('\n') if re.match(r'^_ZN4pgrf3p32\d+render_pass_kernelILi%sELi%sELb0ELb0E.*:' % (KT, V), l)][0].split(':')[0]
body = txt[txt.index('\n' + sym + ':'):]
body = body[:body.index('.end_amdhsa_kernel')]
stages = collections.OrderedDict()
cur = []
for l in body.split('\n'):
    m = re.search(r'; PGRF_STAGE_END (\d+)', l)
    if m:
        stages.setdefault(int(m.group(1)), []).extend(cur); cur = []
        continue
    m = re.match(r'^\s+([vs]_\S+|ds_\S+|global_\S+|buffer_\S+|scratch_\S+)', l)
    if m:
        cur.append(re.sub(r'_e(32|64)$', '', m.group(1)))
stages['tail'] = cur
def cost(op):
    if op in COST: return COST[op]
    if op.startswith('v_'): return SLOW
    if op.startswith('s_waitcnt') or op.startswith('s_barrier'): return 0
    if op.startswith('s_'): return 1.0
    return 8.0
tot = sum(cost(o) for ins in stages.values() for o in ins)
print('%-30s %7s %6s %6s %6s %6s %6s   top non-MFMA mnemonics (cycles)' % ('stage', 'cycles', 'share', 'mfma', 'valu', 'trans', 'mem'))
for k, ins in stages.items():
    c = collections.Counter(); n = collections.Counter()
    for o in ins: c[o] += cost(o); n[o] += 1
    t = sum(c.values())
    mf = sum(v for o, v in c.items() if 'mfma' in o)
    tr = sum(v for o, v in c.items() if cost(o) == TRANS)
    mem = sum(v for o, v in c.items() if o.startswith(('ds_', 'global_', 'buffer_', 'scratch_')))
    va = sum(v for o, v in c.items() if o.startswith('v_') and 'mfma' not in o) - tr
    top = ', '.join('%s %d' % (o.replace('_f32', '').replace('_b32', ''), v) for o, v in c.most_common(7) if 'mfma' not in o)
    print('%-30s %7.0f %5.1f%% %6.0f %6.0f %6.0f %6.0f   %s' % (names.get(k, str(k)), t, 100 * t / tot, mf, va, tr, mem, top))
print('total modelled cycles per 32-point tile: %.0f' % tot)
