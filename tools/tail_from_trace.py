"""round 6: what the ends of a K-step run cost, read off a rocprofv3 kernel trace of `bench.py --timed-only --steps K`: when the last dense launch (k_solve_lean_cl64w4) of the timed
region ends, when the last straggler launch (k_solve_lean_cl4h) ends, and how many dense launches overlap over time.
usage: python tools/tail_from_trace.py <kernel_trace.csv> [K]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
name = lambda r: r.get("Kernel_Name", r.get("Name", ""))
start = lambda r: int(r.get("Start_Timestamp", r.get("BeginNs", 0)))
end = lambda r: int(r.get("End_Timestamp", r.get("EndNs", 0)))
dense = sorted((r for r in rows if "k_solve_lean_cl64w4" in name(r)), key=start)[-K - 1:-1]  # (the last one is the determinism check behind the timed region)
t0 = start(dense[0])
t1 = max(end(r) for r in dense)
helped = [r for r in rows if "k_solve_lean_cl4h" in name(r) and t0 <= start(r) <= t1 + 1000]
ms = lambda t: (t - t0) / 1e6
print("timed region: first dense launch starts at 0, %d dense launches, %d straggler launches" % (len(dense), len(helped)))
print("last dense launch STARTS at %.2f ms, last dense launch ENDS at %.2f ms, last straggler launch ends at %.2f ms" % (
    ms(max(start(r) for r in dense)), ms(max(end(r) for r in dense)), ms(max(end(r) for r in helped))))
ends = sorted(ms(end(r)) for r in dense)
print("dense launches end at (ms): " + " ".join("%.1f" % e for e in ends))
hl = sorted((ms(start(r)), ms(end(r))) for r in helped)
print("straggler launches (start - end, ms): " + " ".join("%.1f-%.1f" % se for se in hl))
dur = sorted(ms(end(r)) - ms(start(r)) for r in helped)
print("straggler launch durations: median %.2f ms, longest %.2f ms" % (dur[len(dur) // 2], dur[-1]))
