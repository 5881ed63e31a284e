#!/usr/bin/env python
"""Assemble the judged summaries under profiles/ from what tools/round_profiles.sh left in gpurun_out/.

    python tools/write_profiles.py <run-tag> <profile-prefix>      e.g.  r01e r01_e
Writes  profiles/<prefix>_{bf16,f32,folded}_kernel_stats.md, <prefix>_pmc_summary.md, <prefix>_bench_{bf16,f32}.json,
<prefix>_stress_gcn.json and refreshes profiles/pmc_traffic.json (read by bench.py for roofline.traffic)."""
import json
import os
import shutil
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GO = os.path.join(ROOT, "gpurun_out")
PR = os.environ.get("MGNNS_PROFILES_OUT") or os.path.join(ROOT, "profiles")     # on the GPU box: a directory under gpurun_out/ (only that is merged back)
os.makedirs(PR, exist_ok=True)


def tables(cur):
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    return lambda p: [x for x in tabs if x.startswith(p)][0]


def kernel_stats(db):
    cur = sqlite3.connect(db).cursor()
    t = tables(cur)
    disp, sym = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    rows = cur.execute("select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), "
                       "max(d.end - d.start) from %s d join %s s on d.kernel_id = s.id group by 1 order by 3 desc"
                       % (disp, sym)).fetchall()
    total = sum(r[2] for r in rows)
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, n, tot, avg, mn, mx in rows:
        short = name if len(name) < 90 else name[:87] + "..."
        lines.append("| `%s` | %d | %.3f | %.2f | %.2f | %.2f | %.1f |" % (short, n, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    # the attention core runs on two problem sizes (image banks L=196 / text bank L=100) with one grid: split at the median
    split = {}
    # (round 4: in bf16 mode the masked text-bank launches run sq_mha32_packed_kernel -- every sq_mha_core_bf16_kernel launch is L=196)
    packed = any("sq_mha32_packed" in r[0] for r in rows)
    for sub in ("sq_mha_core_bf16_kernel", "sq_mha_core_kernel", "sq_mha_core_split_kernel", "folded_attn_kernel", "folded_attn_bf16_kernel"):
        if packed and sub == "sq_mha_core_bf16_kernel":
            continue
        v = sorted(r[0] for r in cur.execute("select d.end - d.start from %s d join %s s on d.kernel_id = s.id where s.kernel_name like ?"
                                             % (disp, sym), ("%" + sub + "%",)))
        if len(v) >= 2:
            h = len(v) // 2
            split[sub] = (sum(v[:h]) / h / 1e3, sum(v[h:]) / (len(v) - h) / 1e3, len(v))
    return "\n".join(lines), split


def pmc(db, sub):
    cur = sqlite3.connect(db).cursor()
    t = tables(cur)
    q = ("select sum(p.value) from %s p join %s d on p.event_id = d.event_id join %s s on d.kernel_id = s.id "
         "where s.kernel_name like ? group by d.event_id order by 1" % (t("rocpd_pmc_event"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")))
    v = [r[0] for r in cur.execute(q, ("%" + sub + "%",))]
    return v


def main():
    tag, pre = sys.argv[1], sys.argv[2]
    # the library that ran must be the one these sources build: a profile is never filed under kernels newer than the ones measured
    sys.path.insert(0, ROOT)
    from mgnns_amd import _lib
    fp = _lib.check_sources("tools/write_profiles.py")
    stamp = "Library sources: fingerprint `%s` (`python -m mgnns_amd.build --fingerprint` on the commit that holds this file).\n\n" % fp
    cmds = {"bf16": "--steps 20 --warmup 5 --no-cpu-baseline --no-variants",
            "bf16_serial": "--steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-graph --single-stream",
            "f32": "--steps 10 --warmup 3 --no-cpu-baseline --no-variants --dtype f32",
            "folded": "--steps 20 --warmup 5 --no-cpu-baseline --no-variants --attn folded",
            "bf16x3_serial": "--steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-graph --single-stream --dtype bf16x3 --attn faithful"}
    for m, args in cmds.items():
        db = os.path.join(GO, "prof_%s_%s" % (tag, m), "%s_%s_results.db" % (tag, m))
        if not os.path.exists(db):
            continue
        table, split = kernel_stats(db)
        log = open(os.path.join(GO, "prof_%s_%s.log" % (tag, m))).read().strip().splitlines()
        line = [x for x in log if x.startswith('{"metric"')]
        with open(os.path.join(PR, "%s_%s_kernel_stats.md" % (pre, m)), "w") as f:
            f.write("# rocprofv3 --kernel-trace --stats, %s, mode %s\n\n" % (pre, m))
            f.write(stamp)
            f.write("Command: `rocprofv3 --kernel-trace --stats -- python3 bench.py %s` (hipGraph replays + the eager timing "
                    "forwards; in the graph replays kernels of the four streams overlap, so a kernel's duration there includes "
                    "what it shares the chip with -- the `bf16_serial` file has every kernel alone on one stream, which is the "
                    "setting bench.py's roofline leg times; summarised from the rocpd database by tools/write_profiles.py).\n\n" % args)
            if line:
                f.write("bench line under the profiler: `%s`\n\n" % line[-1][:400])
            for sub, (lo, hi, n) in split.items():
                f.write("`%s`: %d launches on two problem sizes -- text-bank launches (L=100, masked) mean %.2f us, "
                        "image-bank launches (L=196) mean **%.2f us**.\n\n" % (sub, n, lo, hi))
            f.write(table + "\n")
    # ---- PMC summary -------------------------------------------------------------------------------------------
    rows = []
    traffic = {}
    def half(v, upper):
        h = len(v) // 2
        w = v[h:] if upper else v[:h]
        return sum(w) / max(len(w), 1)
    dbs = {c: os.path.join(GO, "pmc_%s_%s" % (tag, c), "%s_%s_results.db" % (tag, c)) for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES")}
    if all(os.path.exists(p) for p in dbs.values()):
        has_packed = bool(pmc(dbs["FETCH_SIZE"], "sq_mha32_packed"))
        kern = [("sq_mha_core_bf16_kernel L=196 (image banks)", "sq_mha_core_bf16_kernel", None if has_packed else True, "32.1 bank + 1.3 W + 1.0 q + 1.0 out"),
                ("sq_mha_core_bf16_kernel L=100 (text bank, masked)", "sq_mha_core_bf16_kernel_NOT_RUN" if has_packed else "sq_mha_core_bf16_kernel", False, "16.4 bank + 1.3 W + 1.0 q + 1.0 out"),
                ("sq_mha32_packed_kernel L=100 (text bank, masked: live rows packed)", "sq_mha32_packed_kernel", None, "~2.7 live rows (x4 head pairs from L2) + 1.2 W per head pair + 1.0 q + 1.0 out"),
                ("sq_mha32_plan_kernel", "sq_mha32_plan_kernel", None, "0.1 mask"),
                ("folded_attn_bf16_kernel L=196 (image banks)", "folded_attn_bf16_kernel", True, "32.1 bank + 2.5 u + 1.2 c"),
                ("folded_attn_bf16_kernel L=100 (text bank, masked)", "folded_attn_bf16_kernel", False, "<= 16.4 bank (live row tiles only) + 2.5 u + 1.2 c"),
                ("mha_tail_c16", "mha_tail_c16", None, "1.2 c + 3.3 W (L2) + exchange"),
                ("imgbank_pool_bf16", "imgbank_pool_bf16", None, "411.0 map + 1.2 W + 32.1 bank"),
                ("lstm_rec", "lstm_rec", None, "Gx + h"),
                ("mha_tail_bf16", "mha_tail_bf16", None, "1.6 W per WG (L2)"),
                ("textgcn", "textgcn", None, "~5")]
        for label, sub, upper, alg in kern:
            f, w = pmc(dbs["FETCH_SIZE"], sub), pmc(dbs["WRITE_SIZE"], sub)
            if not f:
                continue
            fv = half(f, upper) if upper is not None else sum(f) / len(f)
            wv = half(w, upper) if upper is not None else sum(w) / len(w)
            n = len(f) // 2 if upper is not None else len(f)
            rows.append("| %s | %d | %.0f | %.1f | %.0f | %.1f | %s |" % (label, n, fv, 2 * fv * 1024 / 1e6, wv, wv * 1024 / 1e6, alg))
            if label.startswith("sq_mha_core_bf16_kernel L=196"):
                traffic["sq_mha_core_bf16_kernel@L196"] = {
                    "fetch_kib_raw": fv, "write_kib": wv, "hbm_bytes": int(2 * fv * 1024 + wv * 1024),
                    "source": "profiles/%s_pmc_summary.md: 2 x FETCH_SIZE + WRITE_SIZE (rocprofv3 --pmc, separate passes)" % pre,
                    "kernel_source": "mgnns_amd/csrc/sq_mha_bf16.hip",
                    "kernel_source_sha16": __import__("hashlib").sha256(open(os.path.join(ROOT, "mgnns_amd/csrc/sq_mha_bf16.hip"), "rb").read()).hexdigest()[:16]}
            if label.startswith("folded_attn_bf16_kernel L=196"):
                traffic["folded_attn_bf16@L196"] = {
                    "fetch_kib_raw": fv, "write_kib": wv, "hbm_bytes": int(2 * fv * 1024 + wv * 1024),
                    "source": "profiles/%s_pmc_summary.md: 2 x FETCH_SIZE + WRITE_SIZE (rocprofv3 --pmc, separate passes)" % pre,
                    "kernel_source": "mgnns_amd/csrc/sq_mha_folded_bf16.hip",
                    "kernel_source_sha16": __import__("hashlib").sha256(open(os.path.join(ROOT, "mgnns_amd/csrc/sq_mha_folded_bf16.hip"), "rb").read()).hexdigest()[:16]}
            if label == "imgbank_pool_bf16":
                traffic["imgbank_pool_bf16@B256"] = {
                    "fetch_kib_raw": fv, "write_kib": wv, "hbm_bytes": int(2 * fv * 1024 + wv * 1024),
                    "source": "profiles/%s_pmc_summary.md: 2 x FETCH_SIZE + WRITE_SIZE (rocprofv3 --pmc, separate passes)" % pre,
                    "kernel_source": "mgnns_amd/csrc/imgbank_bf16.hip",
                    "kernel_source_sha16": __import__("hashlib").sha256(open(os.path.join(ROOT, "mgnns_amd/csrc/imgbank_bf16.hip"), "rb").read()).hexdigest()[:16]}
        busy = pmc(dbs["SQ_VALU_MFMA_BUSY_CYCLES"], "sq_mha_core_bf16_kernel")
        with open(os.path.join(PR, "%s_pmc_summary.md" % pre), "w") as f:
            f.write("# rocprofv3 PMC passes, %s (bf16 mode, eager single forwards: `bench.py --steps 3 --warmup 1 --no-cpu-baseline "
                    "--no-variants --no-graph`)\n\n" % pre)
            f.write(stamp)
            f.write("One counter per pass (`rocprofv3 --pmc <ctr> --kernel-trace`, tools/pmc.sh). FETCH_SIZE/WRITE_SIZE are in KiB; per "
                    "MI355X_MICROARCH.md the gfx950 FETCH_SIZE of a wide coalesced read is HALF the bytes (checked on `cast_pad_bf16`: "
                    "15.2 MB raw for a 30.7 MB read), so `hbm_read = 2 x FETCH_SIZE x 1024`; WRITE_SIZE is exact. Values are per launch "
                    "(sum over the counter instances of a dispatch, mean over launches).\n\n")
            f.write("| kernel | launches | FETCH_SIZE raw KiB | read MB (x2) | WRITE_SIZE KiB | write MB | algorithmic MB |\n|---|---|---|---|---|---|---|\n")
            f.write("\n".join(rows) + "\n\n")
            if busy:
                hi = sum(busy) / len(busy) if has_packed else half(busy, True)
                f.write("MFMA pipe: `SQ_VALU_MFMA_BUSY_CYCLES` = %.4e per L=196 launch of sq_mha_core_bf16 (= 256 WG x 8 waves x 2080 MFMA x "
                        "16 cycles: every issued 16x16x32 MFMA, padding included). Divide by 1024 SIMDs x launch time x clock for the pipe "
                        "occupancy; algorithmic utilisation (61.86 GFLOP / time / 2.5 PF) is what bench.py reports.\n" % hi)
        # the split-bf16 attention core (bf16x3 mode): its own two passes
        dx = {c: os.path.join(GO, "pmc_%s_x3_%s" % (tag, c), "%s_x3_%s_results.db" % (tag, c)) for c in ("FETCH_SIZE", "WRITE_SIZE")}
        if all(os.path.exists(p) for p in dx.values()):
            f3, w3 = pmc(dx["FETCH_SIZE"], "sq_mha_core_split_kernel"), pmc(dx["WRITE_SIZE"], "sq_mha_core_split_kernel")
            if f3 and w3:
                fv, wv = half(f3, True), half(w3, True)          # the upper half of the launches by traffic = the L = 196 image banks
                with open(os.path.join(PR, "%s_pmc_summary.md" % pre), "a") as f:
                    f.write("\nbf16x3 mode (`--dtype bf16x3 --attn faithful`), same passes:\n\n| kernel | launches | FETCH_SIZE raw KiB | read MB (x2) | "
                            "WRITE_SIZE KiB | write MB | algorithmic MB |\n|---|---|---|---|---|---|---|\n")
                    f.write("| sq_mha_core_split_kernel L=196 (image banks, hi + lo images) | %d | %.0f | %.1f | %.0f | %.1f | "
                            "64.2 bank (hi + lo) + 2.6 W (hi + lo) + 1.0 q + 1.0 out |\n" % (len(f3) // 2, fv, 2 * fv * 1024 / 1e6, wv, wv * 1024 / 1e6))
                traffic["sq_mha_core_split_kernel@L196"] = {
                    "fetch_kib_raw": fv, "write_kib": wv, "hbm_bytes": int(2 * fv * 1024 + wv * 1024),
                    "source": "profiles/%s_pmc_summary.md: 2 x FETCH_SIZE + WRITE_SIZE (rocprofv3 --pmc, separate passes, bf16x3 mode)" % pre,
                    "kernel_source": "mgnns_amd/csrc/sq_mha_split_bf16.hip",
                    "kernel_source_sha16": __import__("hashlib").sha256(open(os.path.join(ROOT, "mgnns_amd/csrc/sq_mha_split_bf16.hip"), "rb").read()).hexdigest()[:16]}
        if traffic:
            with open(os.path.join(PR, "pmc_traffic.json"), "w") as f:
                json.dump(traffic, f, indent=1)
    # ---- configs[4] SpMM: PMC traffic vs algorithmic bytes --------------------------------------------------------
    spmm_rows = []
    for case, label, alg in (("spmm_d4e-4_F1024", "spmm_bf16 ring kernel, N=10000, density 4e-4, F=1024", 40036 * 6 + 2 * 10000 * 1024 * 2),
                             ("spmm_d4e-4_F2048", "spmm_bf16 register kernel (2 slabs per pass), density 4e-4, F=2048", 40036 * 6 + 2 * 10000 * 2048 * 2),
                             ("spmm_d1e-2_F1024", "spmm_bf16 LDS-tiled kernel, density 1e-2, F=1024", 994841 * 6 + 2 * 10000 * 1024 * 2)):
        vals = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum_TCC_MISS_sum"):
            d = os.path.join(GO, "pmc_%s_%s_%s" % (tag, case, ctr))
            dbs2 = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith(".db")] if os.path.isdir(d) else []
            if dbs2:
                cur = sqlite3.connect(dbs2[0]).cursor()
                t = tables(cur)
                q = ("select i.name, sum(p.value) from %s p join %s d on p.event_id = d.event_id join %s s on d.kernel_id = s.id join %s i on "
                     "p.pmc_id = i.id where s.kernel_name like '%%spmm_bf16%%' group by i.name, d.event_id"
                     % (t("rocpd_pmc_event"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol"), t("rocpd_info_pmc")))
                acc = {}
                for nm, v in cur.execute(q):
                    acc.setdefault(nm, []).append(v)
                for nm, v in acc.items():
                    vals[nm] = sum(v) / len(v)
        kt = os.path.join(GO, "kt_%s_%s" % (tag, case))
        dur = None
        dbs2 = [os.path.join(dp, f) for dp, _, fs in os.walk(kt) for f in fs if f.endswith(".db")] if os.path.isdir(kt) else []
        if dbs2:
            cur = sqlite3.connect(dbs2[0]).cursor()
            t = tables(cur)
            r = cur.execute("select avg(d.end - d.start), count(*) from %s d join %s s on d.kernel_id = s.id where s.kernel_name like '%%spmm_bf16%%'"
                            % (t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol"))).fetchone()
            dur = (r[0] / 1e3, r[1]) if r and r[0] else None
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            rd, wr = 2 * vals["FETCH_SIZE"] * 1024 / 1e6, vals["WRITE_SIZE"] * 1024 / 1e6
            hit = vals.get("TCC_HIT_sum"), vals.get("TCC_MISS_sum")
            spmm_rows.append("| %s | %.1f | %.1f | %.1f | %.2f | %s | %s |" % (
                label, rd, wr, alg / 1e6, (rd + wr) / (alg / 1e6),
                "%.0f / %.0f" % hit if hit[0] is not None else "-", "%.2f us x %d" % dur if dur else "-"))
    if spmm_rows:
        with open(os.path.join(PR, "%s_spmm_pmc.md" % pre), "w") as f:
            f.write("# configs[4] sparse propagation, rocprofv3 PMC passes (%s)\n\n" % pre)
            f.write("`tools/dev/spmm_pmc.sh`: separate `--pmc` passes (FETCH_SIZE, WRITE_SIZE, TCC_HIT_sum + TCC_MISS_sum) + one `--kernel-trace --stats` "
                    "pass of `tools/dev/spmm_prof.py` (cache-cold: every launch takes the next of >= 9 operand sets totalling > 640 MiB). "
                    "FETCH_SIZE doubled per the gfx950 correction (MI355X_MICROARCH.md); values per launch.\n\n")
            f.write("| kernel / case | HBM read MB (2 x FETCH) | write MB | algorithmic MB | traffic / algorithmic | L2 hits / misses per launch | rocprofv3 duration |\n|---|---|---|---|---|---|---|\n")
            f.write("\n".join(spmm_rows) + "\n")
    for name in ("slabcopy", "gather"):
        src = os.path.join(GO, "%s_%s.jsonl" % (tag, name))
        if os.path.exists(src) and os.path.getsize(src):
            shutil.copy(src, os.path.join(PR, "%s_%s.jsonl" % (pre, name)))
    for name in ("bench_bf16", "bench_f32", "bench_bf16x3", "stress_gcn", "bench_bf16_detail", "bench_f32_detail", "bench_bf16x3_detail"):
        src = os.path.join(GO, "%s_%s.json" % (tag, name))
        if os.path.exists(src):
            shutil.copy(src, os.path.join(PR, "%s_%s.json" % (pre, name)))
    print("wrote", sorted(x for x in os.listdir(PR) if x.startswith(pre)))


if __name__ == "__main__":
    main()
