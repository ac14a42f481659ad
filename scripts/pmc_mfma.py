"""gpurun_out/pmc_mfma/*_counter_collection.csv + *_kernel_trace.csv -> profiles/<ROUND>_mfma_util.json: per kernel of the bench step
the share of cycles its SIMDs' matrix pipes were busy.
  SQ_VALU_MFMA_BUSY_CYCLES: cycles a SIMD's MFMA pipe is busy, summed over the SIMDs (32 per v_mfma_f32_32x32x16_f16; MI355X_MICROARCH.md)
  GRBM_GUI_ACTIVE: GPU-busy cycles, reported as the sum over the 8 XCDs -> clock cycles of the dispatch = GRBM_GUI_ACTIVE / 8
  utilisation = MFMA_BUSY / (n_SIMD * GRBM_GUI_ACTIVE / 8), n_SIMD = 256 CUs x 4; effective clock = GRBM_GUI_ACTIVE / 8 / duration.
Counter runs serialise the dispatches (no two-stream overlap), so durations here are those of kernels running ALONE."""
import collections, csv, glob, json, os
ROUND = os.environ.get('ROUND', 'r04')
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = os.path.join(REPO, 'gpurun_out', 'pmc_mfma')
cc = glob.glob(d + '/**/*_counter_collection.csv', recursive=True)[0]
kt = glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0]
dur = {r['Dispatch_Id']: int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(kt))}
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(cc)):
  key = (r['Kernel_Name'].split('(')[0][:70], r['Grid_Size'])
  acc[key][r['Counter_Name']] += float(r['Counter_Value'])
  if (r['Dispatch_Id'], 'n') not in seen:
    seen.add((r['Dispatch_Id'], 'n'))
    cnt[key] += 1
    acc[key]['ns'] += dur.get(r['Dispatch_Id'], 0)
rows = []
for key, c in acc.items():
  n = cnt[key]
  gui = c.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
  if n == 0 or gui <= 0:
    continue
  rows.append({'kernel': key[0], 'grid_threads': int(key[1]), 'launches': n, 'mean_us_alone': c['ns'] / n / 1e3,
               'mfma_util': c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (1024.0 * gui), 'effective_clock_GHz': gui / max(c['ns'], 1.0),
               'total_ms': c['ns'] / 1e6})
rows.sort(key=lambda r: -r['total_ms'])
out = {'source': 'rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE over bench.py '
                 '(scripts/pmc_mfma.sh); mfma_util = MFMA busy cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); counter passes serialise the dispatches',
       'kernels': rows[:24]}
json.dump(out, open(os.path.join(REPO, 'profiles', f'{ROUND}_mfma_util.json'), 'w'), indent=1)
for r in rows[:16]:
  print(f"{r['kernel'][:60]:60s} grid {r['grid_threads']:8d} n {r['launches']:4d}  {r['mean_us_alone']:7.1f} us  MFMA busy {100 * r['mfma_util']:5.1f} %  clock {r['effective_clock_GHz']:.2f} GHz")
