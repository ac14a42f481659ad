"""Diagnostic (needs `make -B EXTRA=-DHALO_STAMP`): in-kernel cycle sums of the 3x3 halo kernel, per wave:
[s_waitcnt + s_barrier in front of each kernel-row group], [group body: fragment reads + 48 MFMAs + DMA / halo issue],
[chunk top: barrier + halo ds_writes], [whole main loop].  Prints the mean over the main-tile workgroups."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from foundationpose_amd import _lib
from foundationpose_amd._lib import check, lib, ptr, stream_ptr
from scripts.bench_conv import SHAPES

def main():
  ctx = _lib.Context.get('cuda:0')
  L = lib()
  for name in sys.argv[1:] or ['encA_res_128', 'encAB_res_256', 'encAB_res_512']:
    N, H, W, Cin, Cout, k, stride, use_res = SHAPES[name]
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn((N, H, W, Cin), device='cuda', generator=g).relu().half()
    w = (torch.randn((Cout, 9 * Cin), device='cuda', generator=g) * (2.0 / (9 * Cin)) ** 0.5).half()
    b = torch.randn((Cout,), device='cuda', generator=g) * 0.1
    res = torch.randn((N, H, W, Cout), device='cuda', generator=g).half()
    out = torch.empty((N, H, W, Cout), device='cuda', dtype=torch.float16)
    run = lambda: check(L.fp_conv2d_f16(ctx.handle, ptr(x), N, H, W, Cin, ptr(w), ptr(b), Cout, 3, 3, 1, 1, ptr(res), 1, ptr(out), 0, stream_ptr()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    L.fp_dbg_halo_stamps(None, 1)
    run(); torch.cuda.synchronize()
    buf = np.zeros((4096, 8, 8), dtype=np.uint64)
    L.fp_dbg_halo_stamps(buf.ctypes.data_as(ctypes.c_void_p), 0)
    nw = 4 if (os.environ.get('FP_HALO_FORM', '0') == '1' and os.environ.get('FP_HALO_NPW', '4') == '2') else 8
    if int(os.environ.get('FP_HALO_DBG', '0')) & 4:
      b = buf[:200, :(4 if (os.environ.get('FP_HALO_FORM', '0') == '1' and os.environ.get('FP_HALO_NPW', '4') == '2') else 8)].reshape(-1, 8)
      b = b[b[:, 7] > 0]
      f = lambda x: float(np.mean(x.astype(np.float64)))
      lo = np.uint64(0xffffffff)
      print(f'{name}: epilogue phases (cycles): issue res/bias loads {f(b[:,0] >> np.uint64(32)):.0f} | barrier {f(b[:,0] & lo):.0f} | stage res + barrier {f(b[:,1] >> np.uint64(32)):.0f} | '
            f'acc->stage {f(b[:,1] & lo):.0f} | barrier {f(b[:,2] & lo):.0f} | total epilogue {f(b[:,5]):.0f}')
      continue
    hw = (buf[:, 0, 2] >> np.uint64(32)).astype(np.int64)
    buf[:, :, 2] &= np.uint64(0xffffffff)
    v = buf[:200, :nw].astype(np.float64)
    v = v[v[..., 3] > 0].reshape(-1, 8)
    groups = 3 * Cin // 32
    m = v.mean(0)
    life = (v[:, 7] - v[:, 6]) * 10.0      # ns (s_memrealtime = 100 MHz)
    print(f'{name}: per group  wait+barrier {m[0]/groups:7.0f}  body {m[1]/groups:7.0f}  | per chunk top {m[2]/(Cin//32):7.0f} | '
          f'prologue {m[4]:7.0f}  loop {m[3]:8.0f}  epilogue {m[5]:7.0f} cycles | lifetime {life.mean()/1e3:6.1f} us -> {(m[3]+m[4]+m[5])/life.mean():.2f} GHz')
    # per-CU timeline: HW_ID bits: wave_id[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13] ... xcc? (print raw layout stats)
    ok = buf[:, 0, 7] > 0
    cu_key = (hw[ok] >> 8) & 0xffffff      # everything above the SIMD field identifies the CU (incl. SE / XCC bits if present)
    ent = (buf[ok, 0, 6].astype(np.float64)); ext = buf[ok, 0, 7].astype(np.float64)
    t00 = ent.min(); ent = (ent - t00) / 100; ext = (ext - t00) / 100
    keys = np.unique(cu_key)
    busy = []; gaps = []
    for kk in keys:
      sel = cu_key == kk
      iv = sorted(zip(ent[sel], ext[sel]))
      busy.append(sum(b - a for a, b in iv))
      gaps.append(len(iv))
    print(f'    distinct CU keys {len(keys)}; workgroups per key min/mean/max {min(gaps)}/{np.mean(gaps):.1f}/{max(gaps)}; '
          f'busy us per key min/mean/max {min(busy):.0f}/{np.mean(busy):.0f}/{max(busy):.0f}; kernel span {ext.max():.0f} us; '
          f'lifetime us p10/p50/p90 {np.percentile(ext-ent,10):.0f}/{np.percentile(ext-ent,50):.0f}/{np.percentile(ext-ent,90):.0f}')
    allv = buf[:, 0, :].astype(np.float64)
    allv = allv[allv[:, 7] > 0]
    t0 = allv[:, 6].min()
    print(f'    workgroups stamped {len(allv)}: first entry 0, last exit {(allv[:, 7].max() - t0) / 100:.1f} us; entry times (us) of blocks 0,256,512,768,1024: '
          + ' '.join(f'{(buf[b, 0, 6] - t0) / 100:.1f}' for b in (0, 256, 512, 768, 1024) if buf[b, 0, 7] > 0))

if __name__ == '__main__':
  main()
