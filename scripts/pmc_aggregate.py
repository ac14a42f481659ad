"""gpurun_out/pmc_traffic/{fetch,write}/*_counter_collection.csv  ->  profiles/<ROUND>_pmc_hbm_traffic.csv, <ROUND>_halo_traffic.json and
<ROUND>_head_chain_traffic.json (ROUND from the environment, default r02).

FETCH_SIZE / WRITE_SIZE are reported in KB.  On gfx950 FETCH_SIZE counts a 128-byte request of a 16-B/lane stream as 64 B:
it is doubled here (MI355X_MICROARCH.md); WRITE_SIZE is taken as is.  Infinity-Cache hits are part of FETCH_SIZE, so the sum
is fabric (L2-miss) traffic - an upper bound on HBM bytes."""
import collections, csv, glob, json, os
ROUND = os.environ.get('ROUND', 'r04')
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(kind, counter):
  f = glob.glob(os.path.join(REPO, 'gpurun_out', 'pmc_traffic', kind, '*_counter_collection.csv'))[0]
  agg = collections.defaultdict(list)
  for r in csv.DictReader(open(f)):
    if r['Counter_Name'] == counter:
      agg[(r['Kernel_Name'], r['Grid_Size'])].append(float(r['Counter_Value']))
  return agg


HEAD_KERNELS = ('tok_gemm', 'tok_qkv', 'head_mlp', 'attention_kernel', 'mean_head', 'vt_pad_zero', 'layernorm', 'ln_partial', 'token_mean', 'conv_igemm2_kernelILi128ELi1', 'conv_igemm2_kernelILi64ELi1')


def chain_bytes(rows):
  """fabric bytes per bench step of the transformer-head kernels: rows = (kernel, grid, launches, raw, fetch MB, write MB)"""
  steps = sum(int(r[2]) for r in rows if 'render_kernel<1' in r[0]) / 6.0        # 6 fused renders per bench step
  tot = collections.OrderedDict()
  for r in rows:
    if any(k in r[0] for k in HEAD_KERNELS):
      key = r[0].split('<')[0].replace('void ', '')[:40]
      tot[key] = tot.get(key, 0.0) + (float(r[4]) + float(r[5])) * int(r[2]) * 1e6 / steps
  return tot


def head_chain(rows):
  """Bytes moved by the head chain (in-projections, attention, out-projection, LayerNorms, FFN, token mean) in one bench step of
  11 head passes (2 heads x 5 refine iterations + ScoreNet's attention), this round against the committed round-1 profile."""
  now = chain_bytes(rows)
  before = {}
  prev = {}
  for tag in ('r01', 'r02', 'r03'):
    old = os.path.join(REPO, 'profiles', tag + '_pmc_hbm_traffic.csv')
    if os.path.exists(old) and tag != ROUND:
      prev[tag] = chain_bytes([(r['kernel'], r['grid_threads'], r['launches'], r['FETCH_SIZE_KB_raw_mean'], r['fetch_MB_corrected_x2'], r['WRITE_SIZE_MB_mean'])
                               for r in csv.DictReader(open(old))])
  before = prev.get('r03') or prev.get('r02') or prev.get('r01') or {}
  js = {'unit': 'bytes per bench step (fetch x2-corrected + write), 11 head passes', 'this_round': now, 'this_round_total': sum(now.values()),
        'previous_rounds': {k: {'kernels': v, 'total': sum(v.values())} for k, v in prev.items()},
        'ratio_to_previous_round': (sum(now.values()) / sum(before.values())) if before else None}
  json.dump(js, open(os.path.join(REPO, 'profiles', f'{ROUND}_head_chain_traffic.json'), 'w'), indent=1)
  print(json.dumps(js, indent=1))


def main():
  fetch, write = load('fetch', 'FETCH_SIZE'), load('write', 'WRITE_SIZE')
  rows = []
  for key in sorted(fetch, key=lambda k: -sum(fetch[k])):
    name, grid = key
    fv, wv = fetch[key], write.get(key, [0.0])
    rows.append((name.split('(')[0][:90], grid, len(fv), sum(fv) / len(fv), 2 * sum(fv) / len(fv) / 1e3, sum(wv) / len(wv) / 1e3))
  out = os.path.join(REPO, 'profiles', f'{ROUND}_pmc_hbm_traffic.csv')
  with open(out, 'w') as f:
    f.write('kernel,grid_threads,launches,FETCH_SIZE_KB_raw_mean,fetch_MB_corrected_x2,WRITE_SIZE_MB_mean\n')
    for r in rows:
      f.write('"%s",%s,%d,%.0f,%.1f,%.1f\n' % r)
  halo = [r for r in rows if 'conv3x3_halo' in r[0] or 'conv3x3_s1_band' in r[0]]      # the 3x3 stride-1 class: halo kernel + band kernel (C = 128)
  n = sum(r[2] for r in halo)
  fe = sum(r[4] * r[2] for r in halo) / n * 1e6
  wr = sum(r[5] * r[2] for r in halo) / n * 1e6
  # algorithmic bytes per launch, mean over the 12 stride-1 3x3 layers of a pass at N=252: input + output (+ residual) + weights
  X = 252 * 1600 * 512.0            # bytes of one fp16 activation tensor of the trunk: 206 MB at every resolution
  algo = (4 * (2 * X) + 2 * X        # C=128 (504 images): 4 convs in+out, 2 residual reads
          + 4 * (2 * X) + 2 * X      # C=256
          + 4 * X + X                # C=512 (tensors are X/2)
          + 4 * 9 * (128 * 128 + 256 * 256 + 512 * 512) * 2.0) / 12
  # `algo` is per FULL-BATCH launch (72 per step); since round 3 the trunk runs as two half batches (144 launches per step, half the
  # bytes each, the weights read twice): scale by the launches actually profiled
  steps = sum(r[2] for r in rows if 'render_kernel<1' in r[0]) / 6.0
  per_step = n / steps if steps else 72.0
  algo = (algo - 4 * 9 * (128 * 128 + 256 * 256 + 512 * 512) * 2.0 / 12) * 72.0 / per_step + 4 * 9 * (128 * 128 + 256 * 256 + 512 * 512) * 2.0 / 12
  js = {'per': 'launch (mean over the stride-1 3x3 convolutions of one bench step, N=252; %.0f launches per step)' % per_step, 'unit': 'bytes', 'fetch': fe, 'write': wr,
        'total': fe + wr, 'algorithmic': algo, 'launches_profiled': n, 'launches_per_step': per_step,
        'source': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py (scripts/pmc_traffic.sh, '
                  'profiles/' + ROUND + '_pmc_hbm_traffic.csv); FETCH_SIZE doubled (gfx950 counts 128-B requests of 16-B/lane streams as 64 B), '
                  'WRITE_SIZE as is; Infinity-Cache hits are included in FETCH_SIZE, so this is fabric (L2-miss) traffic, an upper '
                  'bound on HBM bytes'}
  json.dump(js, open(os.path.join(REPO, 'profiles', f'{ROUND}_halo_traffic.json'), 'w'), indent=1)
  head_chain(rows)
  print(open(out).read()[:1500])
  print(json.dumps(js, indent=1)[:600])


if __name__ == '__main__':
  main()
