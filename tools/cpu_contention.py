"""Does the step time of ONE rank move when seven other busy processes share its host cores?  (An 8-GPU node runs eight such
ranks on one host; no 8-GPU node was available to the builder, so the host side is rehearsed on one GPU.)

    python tools/cpu_contention.py [--cores 8] [--steps 6]

Everything - the bench process and seven siblings that spin two Python threads each, the CPU share a rank's launch loop takes -
is confined to the same `--cores` host cores.  Reported: ms per step and host issue time per step, alone and under load, for the
eager step (~2,400 launches from Python) and for the graph replay (bench.py --graph)."""
import argparse, json, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 1 and sys.argv[1] == "--burn":
    def spin():
        x = 0
        while True:
            x = (x * 1103515245 + 12345) & 0x7FFFFFFF
    threading.Thread(target=spin, daemon=True).start()
    spin()

ap = argparse.ArgumentParser()
ap.add_argument("--cores", type=int, default=8)
ap.add_argument("--steps", type=int, default=6)
a = ap.parse_args()
cores = sorted(os.sched_getaffinity(0))[:a.cores]
os.sched_setaffinity(0, cores)                      # children inherit it (no GPU call in this process)


def bench(extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(a.steps), "--warmup", "2", "--no-roofline",
                        "--no-cpu-baseline"] + extra, capture_output=True, text=True)
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    return line["ms_per_step"], line["config"]["host_issue_ms_per_step"]


out = {"cores": len(cores)}
for name, extra in (("eager", []), ("graph", ["--graph"])):
    alone = bench(extra)
    burners = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--burn"]) for _ in range(7)]
    time.sleep(1.0)
    try:
        loaded = bench(extra)
    finally:
        for b in burners:
            b.kill()
            b.wait()
    out[name] = {"alone_ms_per_step": alone[0], "alone_host_issue_ms": alone[1], "with_7_busy_siblings_ms_per_step": loaded[0],
                 "with_7_busy_siblings_host_issue_ms": loaded[1]}
    print(name, out[name], flush=True)
print(json.dumps(out))
