"""Per-step roofline table of one slice from bench.py --dump-steps JSON:
python tools/steps_report.py steps.json [n_rows]"""
import json
import sys

rows = [r for r in json.load(open(sys.argv[1])) if r["ms"] > 0]   # (slice-invariant steps show 0 ms)
tot = sum(r["ms"] for r in rows)
print("total ms", tot)
P, BW = 157.3e12, 8e12
roof = sum(max(8 * r["macs"] / P, r["bytes"] / BW) for r in rows) * 1e3
print("mixed per-step roofline ms", roof, "-> %.1f%% of it" % (100 * roof / tot))
rows.sort(key=lambda r: -r["ms"])
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    fl = 8 * r["macs"]
    rf = max(fl / P, r["bytes"] / BW) * 1e3
    print(r["step"], r.get("kernel_name", r["kernel"]), "R", r["R"], "K", r["K"], "N", r["N"], "ms=%.2f" % r["ms"],
          "TF=%.1f" % (fl / r["ms"] / 1e9), "GB/s=%.0f" % (r["bytes"] / r["ms"] / 1e6),
          "%.1f%%" % (100 * r["ms"] / tot), "roof=%.2f eff=%.0f%%" % (rf, 100 * rf / r["ms"]))
by = {}
for r in rows:
    d = by.setdefault(r.get("kernel_name", r["kernel"]), [0.0, 0.0, 0])
    d[0] += r["ms"]
    d[1] += max(8 * r["macs"] / P, r["bytes"] / BW) * 1e3
    d[2] += 1
print("by kernel: name, launches, ms, roofline ms, fraction of its roofline")
for k, (ms, rf, n) in sorted(by.items(), key=lambda kv: -kv[1][0]):
    print("  %-52s %4d %9.3f %9.3f %5.0f%%" % (k, n, ms, rf, 100 * rf / ms))
