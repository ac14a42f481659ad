"""Pretty-print the JSON line of bench.py (file argument, else stdin): value, ms/step, roofline, per-class times."""
import json, sys
text = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
d = json.loads([l for l in text.strip().splitlines() if l.startswith('{')][-1])
print(f"{d['value']:.1f} hyp/s  {d['ms_per_step']:.2f} ms/step  halo {d['roofline']['achieved']:.0f} TF/s ({d['roofline']['frac']:.3f})")
for k, v in d.get('kernel_classes', {}).items():
  tf = f"{v['tflops']:.0f} TF/s" if v['tflops'] else ''
  print(f"  {k:14s} {v['ms_per_step']:7.3f} ms/step  {v['launches_per_step']:5.0f} launches  {tf}")
if 'cpu_baseline' in d:
  print('  cpu', d['cpu_baseline'])
