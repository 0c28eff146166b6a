"""Per-step roofline table of one slice from bench.py --dump-steps JSON:
python tools/steps_report.py steps.json [n_rows] [--bf16x3]

Two byte columns per step: ``alg`` = the algorithmic bytes of SURVEY 8(d) (every operand read
once, every result written once PER REFERENCE STEP: a fused pair counts both of its steps, i.e.
also the intermediate it never writes) and ``moved`` = what the plan really moves (a fused pair:
big operand in, result out).  THE BOUND of a step is max(flops / matrix peak, moved bytes / 8 TB/s)
-- a fused pair is priced on the bytes it moves, so no efficiency exceeds 100 %; the unfused
figure is kept as a second column ("unf": what the reference's step-by-step execution would be
bound by).  A stem kernel that multiplies on the bf16 matrix cores (the tenth template argument of
its name; the default arithmetic since round 4) is priced against bf16 peak / 6 products = 416.7
TFLOP/s fp32-equivalent instead of the fp32 pipe's 157.3 (--bf16x3: every stem step, for dumps
without kernel names)."""
import json
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
BF3 = "--bf16x3" in sys.argv
rows = [r for r in json.load(open(args[0])) if r["ms"] > 0]   # (slice-invariant steps show 0 ms)
tot = sum(r["ms"] for r in rows)
P32, PBF3, PH2, BW = 157.3e12, 2500e12 / 6, 2500e12 / 3, 8e12


def peak(r):
    name = r.get("kernel_name") or ""
    if name.startswith("pair_mfma_bf3_kernel"):
        return PBF3
    if name.startswith("pair_mfma_h2_kernel"):
        return PH2
    for prefix, pk in (("stem2h_kernel<", PH2), ("stem2_kernel<", PBF3)):   # (stem2h: fp16 x 2, three products)
        if name.startswith(prefix):
            a = name[len(prefix):].rstrip(">").split(",")
            return pk if len(a) >= 10 and a[9].strip() == "true" else P32
    return PBF3 if (BF3 and r.get("kind") == "stem2") else P32


def bound_ms(r):
    return max(8 * r["macs"] / peak(r), r.get("bytes_moved", r["bytes"]) / BW) * 1e3


def unfused_ms(r):
    return max(8 * r["macs"] / P32, r["bytes"] / BW) * 1e3


print("total ms %.3f" % tot)
roof = sum(bound_ms(r) for r in rows)
unf = sum(unfused_ms(r) for r in rows)
fl = sum(8 * r["macs"] for r in rows)
print("all flops at the fp32 matrix peak (157.3 TFLOP/s) would take %.2f ms: the slice runs at %.1f%% of that rate "
      "(a fraction of a bound only where every step multiplies on the fp32 pipe)" % (fl / P32 * 1e3, 100 * fl / P32 * 1e3 / tot))
print("mixed per-step BOUND (moved bytes) ms %.2f -> the slice runs at %.1f%% of it" % (roof, 100 * roof / tot))
print("mixed per-step roofline of the UNFUSED steps (SURVEY 8d bytes) ms %.2f (the reference's execution model; "
      "a fused pair may finish below it)" % unf)
rows.sort(key=lambda r: -r["ms"])
print("step kernel R K N | ms  share | TFLOP/s  moved GB/s (alg GB/s) | bound ms  eff | unfused roof ms")
for r in rows[: int(args[1]) if len(args) > 1 else 25]:
    f = 8 * r["macs"]
    mv = r.get("bytes_moved", r["bytes"])
    b = bound_ms(r)
    print(r["step"], r.get("kernel_name", r["kernel"]), "R", r["R"], "K", r["K"], "N", r["N"], "| ms=%.2f" % r["ms"],
          "%.1f%%" % (100 * r["ms"] / tot), "| TF=%.1f" % (f / r["ms"] / 1e9), "GB/s=%.0f" % (mv / r["ms"] / 1e6),
          "(alg %.0f)" % (r["bytes"] / r["ms"] / 1e6), "| bound=%.2f eff=%.0f%%" % (b, 100 * b / r["ms"]),
          "| unf=%.2f" % unfused_ms(r))
by = {}
for r in rows:
    d = by.setdefault(r.get("kernel_name", r["kernel"]), [0.0, 0.0, 0, 0.0])
    d[0] += r["ms"]
    d[1] += bound_ms(r)
    d[2] += 1
    d[3] += 8 * r["macs"]
print("by kernel: name, launches, ms, bound ms, fraction of its bound, TFLOP/s")
for k, (ms, rf, n, f) in sorted(by.items(), key=lambda kv: -kv[1][0]):
    print("  %-56s %4d %9.3f %9.3f %5.0f%% %7.1f" % (k, n, ms, rf, 100 * rf / ms, f / ms / 1e9))
