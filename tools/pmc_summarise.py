"""Turns the rocprofv3 CSVs of tools/collect_profiles.sh into the JSON summaries bench.py reads:
profiles/<tag>_pmc_traffic.json (HBM bytes per launch: FETCH_SIZE x2 on gfx950 per /opt/skills/guides/MI355X_MICROARCH.md,
WRITE_SIZE exact; both are reported in KB) and profiles/<tag>_pmc_counters.json (MFMA / VALU / LDS counters per launch and
the per-trajectory-step figures derived from them), plus profiles/<tag>_kernel_stats.csv."""
import csv, re
import glob
import json
import os
import shutil
import sys

out, tag = sys.argv[1], sys.argv[2]
# third argument "per_step": the passes ran `bench.py --streamed-jacobians` (residual Jacobians given per step, the round-1..3 form);
# default: the constant residual Jacobian uploaded once (round 4's bench default)
PER_STEP = len(sys.argv) > 3 and sys.argv[3] == "per_step"
SUFFIX = "_per_step_jacobians" if PER_STEP else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T, B = 3000, 1024

KEYS = {"k_backward_fused": "backward_fused", "k_forward_fused": "forward_fused", "k_fd_difference": "fd_difference"}


def key_of(name):
    # the one-wave forward sweep is launched in two forms (uniform key-point sets / general), one of which returns at
    # once: only the uniform form runs the bench workload (set_interval key-points)
    if re.search(r"k_forward_fused(_excl)?<\d+, \d+, (true|false), false(, (true|false))?>", name):
        return None
    # the same for the one-wave backward sweep since round 3 (<N, M, RU0, RAW, UNI[, RXC]>: UNI = false leaves at once here)
    if re.search(r"k_backward_fused(_excl)?<\d+, \d+, (true|false), (true|false), false(, (true|false))?>", name):
        return None
    for k, v in KEYS.items():
        if k in name:
            return v
    return None


def counters(passdir):
    """-> {kernel key: {counter: mean per launch (summed over the device)}}"""
    acc = {}
    # (a directory that several collection runs were merged into holds one file per run: the NEWEST is the run being summarised)
    for f in sorted(glob.glob(os.path.join(out, passdir, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]:
        per_dispatch = {}
        for row in csv.DictReader(open(f)):
            k = key_of(row["Kernel_Name"])
            if k is None:
                continue
            d = per_dispatch.setdefault((k, row["Dispatch_Id"]), {})
            d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        for (k, _), d in per_dispatch.items():
            for cn, v in d.items():
                acc.setdefault(k, {}).setdefault(cn, []).append(v)
    return {k: {cn: sum(v) / len(v) for cn, v in d.items()} for k, d in acc.items()}, \
           {k: len(next(iter(d.values()))) for k, d in acc.items()}


fetch, nl = counters("pmc_fetch")
write, _ = counters("pmc_write")
traffic = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes over `python bench.py --no-cpu-baseline "
                   "--no-secondary --steps 3 --warmup 1` (Panda reaching, B=1024 distinct seeds, T=3000, fused sweeps, key-point ordered FD payload differenced inside the backward sweep" + (", residual Jacobians per step" if PER_STEP else ", ONE constant residual Jacobian uploaded once") + "); KB per launch, mean over launches. "
                   "gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request -> x2 (calibrated in round 1 on a "
                   "kernel with exactly known reads); WRITE_SIZE is exact.",
           "workload": {"task": "panda_reaching", "T": T, "batch": B, "rx_const": not PER_STEP}, "kernels": {}}
for k in fetch:
    fk, wk = fetch[k].get("FETCH_SIZE", 0.0), write.get(k, {}).get("WRITE_SIZE", 0.0)
    traffic["kernels"][k] = {"FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "traffic_bytes": 1024.0 * (2.0 * fk + wk), "launches": nl.get(k)}
json.dump(traffic, open(os.path.join(root, "profiles", f"{tag}_pmc_traffic{SUFFIX}.json"), "w"), indent=1)

m3, _ = counters("pmc_m3")
m4, _ = counters("pmc_m4")
cnt = {"note": "rocprofv3 --pmc, two separate passes (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU; "
               "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY) over the same command; mean per launch, "
               "summed over the device.  Counters count per wave: / (T x B) gives the per-trajectory-step figures.",
       "derived": {}, "kernels": {}}
for k in m3:
    d = dict(m3[k]); d.update(m4.get(k, {}))
    cnt["kernels"][k] = d
    if d.get("SQ_INSTS_VALU_MFMA_F64"):
        mf = d["SQ_INSTS_VALU_MFMA_F64"]
        cnt["derived"][k] = {"mfma_per_step_per_trajectory": mf / (T * B),
                             "mfma_busy_cycles_per_instruction": d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / mf,
                             "valu_instructions_per_step_per_trajectory": d.get("SQ_INSTS_VALU", 0.0) / (T * B),
                             "lds_bank_conflict_cycles": d.get("SQ_LDS_BANK_CONFLICT", 0.0)}
json.dump(cnt, open(os.path.join(root, "profiles", f"{tag}_pmc_counters{SUFFIX}.json"), "w"), indent=1)

for sub, name in (("stats", f"{tag}_kernel_stats{SUFFIX}.csv"), ("stats_generic", f"{tag}_generic_kernel_stats.csv"),
                  ("stats_reach_velocity_change", f"{tag}_kernel_stats_velocity_change_lists.csv"),
                  ("stats_reach_adaptive_jerk", f"{tag}_kernel_stats_adaptive_jerk_lists.csv"),
                  ("stats_b512", f"{tag}_kernel_stats_b512.csv"), ("stats_b256", f"{tag}_kernel_stats_b256.csv"),
                  ("stats_configs2", f"{tag}_kernel_stats_configs2_pushing_b64.csv"), ("stats_configs4", f"{tag}_kernel_stats_configs4_n62_b128.csv")):
    fs = sorted(glob.glob(os.path.join(out, sub, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if fs:
        shutil.copy(fs[-1], os.path.join(root, "profiles", name))
for src, name in (("bench.json", f"{tag}_bench{SUFFIX}.json"), ("bench_detail.json", f"{tag}_bench_detail{SUFFIX}.json"), ("generic.json", f"{tag}_generic_bench.json")):
    if os.path.exists(os.path.join(out, src)):
        shutil.copy(os.path.join(out, src), os.path.join(root, "profiles", name))
print(json.dumps(cnt["derived"], indent=1))
print({k: v["traffic_bytes"] for k, v in traffic["kernels"].items()})
