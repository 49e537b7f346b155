#!/usr/bin/env python3
"""overlap_probe.py — driver of overlap_probe.hip (see there). Per weight size: us per "projection" in a captured chain
of data-dependent launches,
  serial     one stream, stream order is the dependency (what a decode step does today)
  overlap    two streams alternating, consumer waits on the producer's device-side counter, first tiles requested
             BEFORE the wait
  overlap_np same, first tiles requested AFTER the wait (dispatch overlap only)
Build: hipcc --offload-arch=gfx950 -O3 -fPIC -shared tools/probe/overlap_probe.hip -o tools/probe/liboverlap_probe.so"""
import argparse, ctypes, json, os
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "liboverlap_probe.so"))
P, U64, I32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int32
lib.probe_bump.argtypes = [P, P]
lib.probe_launch.argtypes = [P, U64, I32, P, P, P, I32, P, P, P, I32, I32, P, I32]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chain", type=int, default=32)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--wgs", default="256,512")
    ap.add_argument("--mbs", default="33.5,50.3,117.4,234.9")
    ap.add_argument("--spin", type=int, default=200000)
    ap.add_argument("--threads", default="256")
    ap.add_argument("--modes", default="serial,overlap,overlap_np")
    a = ap.parse_args()
    n = a.chain
    pool = torch.empty(int(3.2e9) // 4, dtype=torch.int32, device="cuda").random_(0, 1 << 30)
    done = torch.zeros(n, dtype=torch.int64, device="cuda")
    step = torch.zeros(1, dtype=torch.int64, device="cuda")
    xbuf = torch.zeros(n + 1, 1024, dtype=torch.int32, device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    for mb in [float(v) for v in a.mbs.split(",")]:
        for wgs, threads in [(int(v), int(t)) for v in a.wgs.split(",") for t in a.threads.split(",")]:
            nbytes = int(mb * 1e6) // (threads * 8 * 16 * wgs) * (threads * 8 * 16 * wgs)
            copies = max(2, int(3.0e9 // nbytes))
            res = {"MB": round(nbytes / 1e6, 1), "wgs": wgs, "threads": threads, "chain": n}
            for mode in a.modes.split(","):
                done.zero_(); step.zero_(); err.zero_()
                torch.cuda.synchronize()
                sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=sA):
                    lib.probe_bump(step.data_ptr(), sA.cuda_stream)
                    if mode != "serial":
                        sB.wait_stream(sA)
                    for i in range(n):
                        w_ptr = pool.data_ptr() + (i % copies) * nbytes
                        if mode == "serial":
                            rc = lib.probe_launch(w_ptr, nbytes, wgs, None, None, step.data_ptr(), wgs,
                                                  xbuf[i].data_ptr(), xbuf[i + 1].data_ptr(), err.data_ptr(), a.spin, 1,
                                                  sA.cuda_stream, threads)
                        else:
                            s = sA if i % 2 == 0 else sB
                            rc = lib.probe_launch(w_ptr, nbytes, wgs, done[i - 1:].data_ptr() if i else None,
                                                  done[i:].data_ptr(), step.data_ptr(), wgs, xbuf[i].data_ptr(),
                                                  xbuf[i + 1].data_ptr(), err.data_ptr(), a.spin,
                                                  1 if mode == "overlap" else 0, s.cuda_stream, threads)
                        assert rc == 0
                    if mode != "serial":
                        sA.wait_stream(sB)
                with torch.cuda.stream(sA):
                    g.replay(); g.replay()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(a.reps):
                        g.replay()
                    e1.record()
                    e1.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / (a.reps * n)
                res[mode + "_us"] = round(us, 2)
                res[mode + "_TBps"] = round(nbytes / us / 1e6, 2)
                if mode != "serial":
                    res[mode + "_err"] = int(err.item())
                    res[mode + "_done_ok"] = bool((done == (a.reps + 2) * wgs).all().item())
                del g
            print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
