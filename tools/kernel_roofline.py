#!/usr/bin/env python3
"""Per-kernel view of a rocprofv3 --kernel-trace --stats run of bench.py at N^3: calls per step, average duration, the
algorithmic bytes DESIGN.md section 4 / SURVEY 8(d) assign to the kernel's family, and what that makes against the 8 TB/s
HBM peak.  The byte figures are per VOXEL of one launch (fields x 4 B): they say how far a kernel is from moving its data
once, not how busy the memory system is (the gather family is bound by VALU issue and the texture-addresser path).

    python tools/kernel_roofline.py profiles/r03_zh_driver_command_kernel_stats.csv --n 256 --steps 25 > profiles/<name>.md
"""
import argparse, csv, re

# (regex on the demangled kernel name, algorithmic bytes per voxel per launch, bound named in DESIGN.md section 4)
FAMILIES = [
    (r"jacobi_lds_kernel<\d+, \d+, 3>|jacobi_lds2seg_kernel", 12, "fabric / HBM; compulsory bytes of a 3-sweep launch"),
    (r"jacobi_lds_kernel<\d+, \d+, 4>", 12, "fabric; compulsory bytes of a 4-sweep launch"),
    (r"jacobi_lean3r_kernel|jacobi_lean2r_kernel|jacobi_march2", 12, "fabric / HBM; compulsory bytes of a fused launch"),
    (r"jacobi_march_kernel|jacobi_generic|jacobi_tile", 12, "fabric / HBM"),
    (r"advect_kernel<.*, 2, true>", 40, "VALU issue + texture addresser (two fields)"),
    (r"advect_kernel", 20, "VALU issue + texture addresser"),
    (r"compensate_kernel<.*, 2, true>", 48, "VALU issue + texture addresser (two fields)"),
    (r"compensate_kernel", 24, "VALU issue + texture addresser"),
    (r"cumulate_kernel<.*, 2, (true|false), (true|false)>", 48, "VALU issue + texture addresser (two fields)"),
    (r"cumulate_kernel", 24, "VALU issue + texture addresser"),
    (r"forward_kernel|dmc_kernel", 36, "latency of three dependent gather rounds"),
    (r"clamp_box", 12, "L2"),
    (r"gradient_delta_kernel", 52, "HBM"),
    (r"gradient_kernel", 28, "HBM"),
    (r"divergence_kernel", 16, "HBM"),
    (r"buoyancy_kernel", 16, "HBM"),
    (r"add_field_kernel", 12, "HBM"),
    (r"max_abs3_partial_kernel", 12, "HBM"),
    (r"init_maps_kernel", 12, "HBM (write only)"),
    (r"copyBuffer", 8, "HBM"),
    (r"fillBuffer", 4, "HBM (write only)"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("stats")
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--steps", type=int, required=True, help="steps the profiled command ran (warm-up included)")
    a = ap.parse_args()
    vox = a.n ** 3
    rows = list(csv.DictReader(open(a.stats)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"| kernel | launches / step | avg µs | ms / step | % | algorithmic B / voxel | GB/s | of 8 TB/s | bound (DESIGN §4) |")
    print("|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        t = float(r["TotalDurationNs"])
        if t / total < 0.002:
            continue
        name = r["Name"].replace("void ", "").replace("bq::exact::", "").replace("bq::", "")
        short = name.split("(")[0]
        fam = next(((b, why) for pat, b, why in FAMILIES if re.search(pat, name)), None)
        avg = float(r["AverageNs"]) / 1e3
        if fam:
            gbs = fam[0] * vox / (avg * 1e-6) / 1e9
            cols = f"{fam[0]} | {gbs:,.0f} | {gbs / 8000:.2f} | {fam[1]}"
        else:
            cols = "— | — | — | —"
        print(f"| `{short[:70]}` | {int(r['Calls']) / a.steps:.1f} | {avg:.1f} | {t / a.steps / 1e6:.3f} | {100 * t / total:.1f} | {cols} |")
    print(f"\nkernel time per step: {total / a.steps / 1e6:.3f} ms ({a.steps} steps, {a.n}^3)")


if __name__ == "__main__":
    main()
