#!/usr/bin/env python3
"""bench.py -- throughput of the k-mer classification hot path on MI355X.

A "step" is `--reps-per-step` passes of the hot path (shk_classify_device:
FastqSplitter join/mask semantics + ReadAnalyzer + BF::get_index, SURVEY.md 8a
rows 13-15) over ONE fixed set of synthetic read pairs already resident in HBM.

Headline workload (BASELINE.json configs[1], the configuration the metric is
quoted on): 1 gene x 20 kb uniform-ACGT reference, synthetic 2x150 bp pairs
(50 % on-target, 1 % substitutions, 0.2 % N), k=17, c=0.6, 2^33-bit filter,
classified in launches of 10 M pairs.

Scaling (--scaling, default strong): the read set of --total-pairs pairs
(default 80 M = 8 chunks of 10 M pairs; chunk c is generated from seed SEED+1+c
whatever the number of GPUs) is split over the ranks, so N GPUs classify the
same reads 1 GPU does and `gene_count_checksum` / `assoc_per_step` must not
change with N.  A step loops a rank's chunks --reps-per-step times (default 16,
identically at every N), so the timed window stays long against launch and
collective latencies on 8 GPUs too: 8 GPUs x 1 chunk x 16 passes x 20 steps is
about 1.4 s.  Every rank holds its own replica of the index (rebuilt
deterministically; build time reported separately, not timed); there is no
data-path collective.

WHAT IS INSIDE THE TIMED WINDOW: barrier + synchronize, K steps of classify
launches (each shk_classify_device call synchronises its stream once), ONE
all-reduce of the per-gene assigned-read counts over RCCL issued by the library
(shk_dist_gene_counts_allreduce), barrier + synchronize.  Outside it: W warm-up
steps and one warm-up all-reduce (RCCL sets its channels up on a communicator's
first collective).  `per_rank` carries every rank's kernel / wall / all-reduce /
barrier-wait split, gathered over the job's own channel after the window.

Prints ONE JSON line on rank 0 (contract in the task statement) carrying
  roofline       against the resource that binds the dominant kernel, in that
                 resource's units, frac <= 1: the headline index sits in LDS and
                 the kernel is bound by VALU issue (achieved = G wave-
                 instructions/s from the SQ counters, peak = 1024 SIMDs x
                 2.4 GHz / 2); the configs[2] / configs[4] indices are bound by
                 the RATE of random memory-side requests (achieved = L2 misses
                 per second, peak = the ceiling measured in this run).
                 `hbm_compulsory` = (input + result bytes) / kernel time against
                 the 8 TB/s HBM peak; `model_8d` = SURVEY 8(d)'s algorithmic-byte
                 figure under its own key (it counts every k-mer of every read as
                 the reference visits them and may exceed the HBM peak: most of
                 those probes are proven irrelevant and never made).  Counters
                 are taken LIVE: rank 0 at N=1 runs rocprofv3 --pmc passes of
                 the same workloads as child processes (same seeds, same
                 library); `counters_source` says so, or names the committed
                 profiles/pmc_counters.json (only used when taken on the same
                 kernel sources) when rocprofv3 is not available
  cpu_baseline   the CPU oracle (a port of the reference path) on this host
  configs        the same measurement on BASELINE configs[2]'s index
                 (60 000 genes, 2^36-bit filter) and on configs[4]'s shape
                 (k=31, -q 20, --single, 2^37 bits), 10 M-pair steps
  batch_boundary PCIe-inclusive rate of the host-buffer entry points (never
                 `value`)
  cli_end_to_end the `shark` command on FASTQ files (never `value`): two sample
                 sizes x two on-target rates; at N>1 the host-fed multi-GPU leg
                 (`shark --gpus N`)
"""
import argparse
import hashlib
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver (already exported on the pool)
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_SIMD = 256 * 4            # 256 CUs x 4 SIMD-32
CLK_GHZ = 2.4               # max clock; a wave64 VALU instruction issues over 2 cycles
VALU_PEAK_GINST = N_SIMD * CLK_GHZ / 2.0   # 1228.8 G wave-instructions/s
LAUNCH_PAIRS = 10_000_000
L2_LINE_BYTES = 128         # a memory-side request of gfx950's L2 is a 128-byte line (MI355X_MICROARCH.md: FETCH_SIZE tallies 128-B requests at 64 B)
KERNEL_SOURCES = ["classify_uni.hpp", "classify_uni_plan.inc", "classify_uni_loads.inc", "classify_uni_staging.inc", "classify_uni_tiles.inc", "classify_uni_rounds.inc",
                  "classify_uni_vote.inc", "classify_uni_anchored.inc", "classify_uni_sparse.inc", "classify_common.hpp", "classify.hip", "anchor_verdict.hip", "kmer_device.hpp", "shark_internal.hpp", "lds_table.hpp"]
# one rocprofv3 --pmc pass per entry (gfx950: 8 SQ slots; FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2)
COUNTER_SETS = [
    ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"],
    # the clock the chip held during the kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the dispatch's duration in the same pass
    # (MI355X_MICROARCH.md, "DVFS give-back"); a set of its own, so that a box without the counter loses nothing else
    ["GRBM_GUI_ACTIVE"],
]
COUNTER_SETS_FULL = COUNTER_SETS + [["SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_VMEM", "SQ_INSTS_VALU"]]
WORKLOADS = ("configs1", "configs2", "configs4_uniform", "configs4_ends")
CAL_TABLE_BYTES = 8 << 30    # the FETCH_SIZE calibration's table (far beyond the 256 MiB Infinity Cache) and its lookups
CAL_LOOKUPS = 1 << 30


_T0 = time.time()


def log(msg):
    """progress on stderr (the JSON line is the only thing on stdout): which phase a run is in, with seconds since start"""
    if os.environ.get("RANK", "0") == "0":
        print("[bench %6.1f s] %s" % (time.time() - _T0, msg), file=sys.stderr, flush=True)


def kernel_src_sha():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "shark_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def cpu_info():
    model, phys = "", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and not model:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    return model, len(phys)


def host_threads():
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def shard_chunks(scaling, total_pairs, launch_pairs, world, rank):
    """which chunks of the read set `rank` of `world` owns -> (pairs per chunk, [chunk ids], pairs of the whole set per pass).
    strong: ONE read set of total_pairs, cut into chunks of about launch_pairs (chunk c is generated from seed SEED+1+c whatever
    the number of GPUs), as many chunks as makes the shards equal; weak: one chunk of launch_pairs per rank."""
    if scaling == "strong":
        n_chunks = max(world, (total_pairs + launch_pairs - 1) // launch_pairs)
        n_chunks = (n_chunks + world - 1) // world * world           # equal shards
        chunk_pairs = total_pairs // n_chunks
        per = n_chunks // world
        return chunk_pairs, list(range(rank * per, (rank + 1) * per)), chunk_pairs * n_chunks
    return launch_pairs, [rank], launch_pairs * world


def spawn_ranks(n):
    """`python3 bench.py --gpus N` without a launcher: this process starts the N ranks itself, one process per GPU -- as the
    reference starts its N workers from the one command line (main.cpp:219-223) -- forwards rank 0's JSON line and exits
    non-zero when any rank does.  It never imports torch and never touches a GPU (a process that has initialised the GPU
    must not be replaced or forked on this pool); the ranks are ordinary child processes of it."""
    import socket
    import subprocess
    import threading
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout.readlines()), daemon=True)
    reader.start()
    rc = 0
    live = set(range(n))
    while live and rc == 0:
        for r in sorted(live):
            c = procs[r].poll()
            if c is not None:
                live.discard(r)
                if c != 0:
                    rc = c if c > 0 else 1
                    print("bench.py: rank %d of %d exited with code %d" % (r, n, c), file=sys.stderr)
        time.sleep(0.05)
    for r in live:          # a rank failed: the others would wait for it at the next barrier -- stop exactly the processes started here
        procs[r].terminate()
    for r in live:
        try:
            procs[r].wait(timeout=20)
        except subprocess.TimeoutExpired:
            procs[r].kill()
    reader.join(timeout=20)
    sys.stdout.write(b"".join(lines).decode(errors="replace"))
    sys.stdout.flush()
    sys.exit(rc)


# =====================================================================================================================
# the workloads (one definition for the timed run, the counter child and tools/gpu_profiles.sh)
# =====================================================================================================================
def workload_spec(name, args):
    """-> dict(k, bf_log2, q, single, genes (callable), on_target, with_qual, qual_model, seed_offset)"""
    from shark_amd import synth
    if name == "configs1":
        return dict(k=args.k, bf_log2=args.bf_log2, q=0, single=False, genes=lambda: synth.make_reference(args.genes, args.gene_len),
                    on_target=args.on_target, with_qual=False, qual_model=None, seed=synth.SEED + 1)
    if name in ("configs2", "configs2_ot1.00"):      # (the second: every pair drawn from a gene -- instructions per ON-TARGET pair, profiles only)
        return dict(k=17, bf_log2=36, q=0, single=False, genes=lambda: synth.make_gencode_like_reference(60000),
                    on_target=0.5 if name == "configs2" else 1.0, with_qual=False, qual_model=None, seed=synth.SEED + 7)
    if name in ("configs4_uniform", "configs4_ends"):
        return dict(k=31, bf_log2=37, q=20, single=True, genes=lambda: synth.make_gencode_like_reference(60000),
                    on_target=0.5, with_qual=True, qual_model=name.split("_")[1], seed=synth.SEED + 7)
    raise SystemExit("bench.py: unknown workload %r" % name)


def counter_child(args):
    """--counter-child W1,W2,...: run under `rocprofv3 --pmc ... --kernel-trace`.  For every workload: build its index, generate its
    10 M-pair batch (the seed the timed run uses for the same workload on rank 0), one warm-up launch and two more.  Prints one
    JSON line {workload: number of shk_classify_device calls, ...} in the order they were made; the parent cuts the counter
    rows into calls at the uniform_check_kernel dispatch that opens each of them."""
    import torch
    from shark_amd import SharkHip, synth
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    L, order, genes_cache = 150, [], {}
    for name in args.counter_child.split(","):
        sp = workload_spec(name, args)
        key = "ref1" if name == "configs1" else "ref60k"
        if key not in genes_cache:
            genes_cache[key] = sp["genes"]()
        genes = genes_cache[key]
        h = SharkHip(k=sp["k"], c=0.6, bf_bits=1 << sp["bf_log2"], min_quality=sp["q"], single=sp["single"], device=0)
        h.build([g.tobytes() for g in genes])
        kw = dict(seed=sp["seed"], read_len=L, on_target=sp["on_target"], with_qual=sp["with_qual"])
        if sp["qual_model"]:
            kw["qual_model"] = sp["qual_model"]
        b = synth.make_pairs_device(args.pairs, genes, dev, **kw)
        torch.cuda.synchronize()
        p = {kk: (v.data_ptr() if v is not None else 0) for kk, v in b.items()}
        calls = 3
        for _ in range(calls):
            r = h.classify_device(args.pairs, p["seq1"], p["off1"], p["seq2"], p["off2"], p["qual1"], p["qual2"], max_read_len=L)
        order.append({"workload": name, "calls": calls, "kernel": h.last_kernel(), "n_assoc": int(r.n_assoc), "probe_mode": h.probe_mode()})
        h.close()
        del b, p
    # the issue ceiling's kernel (shk_measure_valu_mix): its instruction count per wave-iteration comes from this same counter pass
    hm = SharkHip(k=args.k, c=0.6, bf_bits=1 << 20, device=0)
    mix = []
    for wv in (4, 8):
        ms, wi = hm.measure_valu_mix(wv, 2000)
        mix.append({"waves_per_simd": wv, "wave_iterations_timed_launch": wi, "warmup_wave_iterations": wi // 2000 * 64})
    # the calibration of FETCH_SIZE for the position table's access pattern: random 16-byte lookups in an 8 GiB table -- a KNOWN number
    # of lines, once with one read per line, once with a read in each 64-byte half of every line (shk_measure_random_lookups, bit 1)
    cal = {"table_bytes": CAL_TABLE_BYTES, "lines": hm.random_lookups_made(CAL_LOOKUPS)}
    for both in (False, True):
        cal["G_lines_per_s_%s" % ("both_halves" if both else "one_read")] = round(hm.measure_random_lookups(CAL_TABLE_BYTES, CAL_LOOKUPS, True, both), 2)
    hm.close()
    print(json.dumps({"counter_child": order, "valu_mix": mix, "lookup_calibration": cal}), flush=True)


def parse_counter_dir(d, order):
    """rocprofv3's counter_collection CSV of one child -> {workload: {kernel, counters{name: value per launch}}}: dispatches in
    Dispatch_Id order, cut into shk_classify_device calls at every uniform_check_kernel; per call the classify kernel that did the
    work (largest value; the instantiation that returns at once counts ~nothing); per workload the LAST call (the first is warm-up)."""
    import csv
    import glob
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for i, r in enumerate(csv.DictReader(fh)):
                try:
                    did = int(r.get("Dispatch_Id") or r.get("Dispatch_ID") or i)
                except ValueError:
                    did = i
                try:
                    dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                except (KeyError, ValueError, TypeError):
                    dur = None
                rows.append((did, r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"]), dur))
    rows.sort(key=lambda x: x[0])
    calls, cur, last_did = [], None, None
    for did, kn, cn, v, dur in rows:
        if "uniform_check_kernel" in kn:
            if did != last_did:
                cur = {}
                calls.append(cur)
                last_did = did
            continue
        if cur is None or not ("classify_" in kn or "anchor_verdict" in kn) or "classify_general" in kn:
            continue
        name = kn.split("(")[0].replace("void shk::", "")
        cur.setdefault(name, {})
        cur[name][cn] = cur[name].get(cn, 0.0) + v      # (a counter may be reported per XCD / per dimension: summed)
        if cn == "GRBM_GUI_ACTIVE" and dur:
            cur[name]["GRBM_PASS_NS"] = dur            # (the dispatch's duration in the pass that counted it)
    out, at = {}, 0
    for o in order:
        mine = calls[at:at + o["calls"]]
        at += o["calls"]
        if not mine or not mine[-1]:
            continue
        last = mine[-1]
        kn = max(last, key=lambda k: max(last[k].values()) if last[k] else 0.0)
        # a launch's work may be split over two kernels (anchor_verdict_kernel in front of the table kernel): their counters are added --
        # the library's kernel time covers both --, the instantiations that return at once add next to nothing
        tot = {}
        for name, cs in last.items():
            for cn, v in cs.items():
                if cn == "GRBM_PASS_NS":
                    continue
                tot[cn] = tot.get(cn, 0.0) + v
        if "GRBM_PASS_NS" in last[kn]:
            tot["GRBM_GUI_ACTIVE"] = last[kn].get("GRBM_GUI_ACTIVE", 0.0)       # (the clock: of the dominant kernel alone)
            tot["GRBM_PASS_NS"] = last[kn]["GRBM_PASS_NS"]
        parts = {name: {cn: cs[cn] for cn in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "TCC_MISS_sum", "FETCH_SIZE") if cn in cs}
                 for name, cs in last.items() if cs and max(cs.values()) > 0.01 * max(last[kn].values())}
        out[o["workload"]] = {"kernel": kn, "counters": tot, "kernels": parts}
    return out


def parse_kernel_counter(d, kernel, counter):
    """values of one counter for every dispatch of `kernel`, in dispatch order (summed over the rows of a dispatch)"""
    import csv
    import glob
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for i, r in enumerate(csv.DictReader(fh)):
                if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                    try:
                        did = int(r.get("Dispatch_Id") or r.get("Dispatch_ID") or i)
                    except ValueError:
                        did = i
                    acc[did] = acc.get(did, 0.0) + float(r["Counter_Value"])
    return [acc[k] for k in sorted(acc)]


def collect_counters(args, workloads, sets=None, keep_dir=None, timeout_s=420):
    """one rocprofv3 --pmc child per counter set (separate passes, --kernel-trace only beside --pmc) over `workloads`
    -> ({workload: {"kernel":..., counter: value per launch, ...}}, note).  Never raises: a bench line without live counters
    says why in `note`."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return {}, "rocprofv3 not found"
    res, notes = {w: {} for w in workloads}, []
    base = keep_dir or tempfile.mkdtemp(prefix="shark_pmc_", dir="/tmp")
    t0 = time.time()
    try:
        for si, cset in enumerate(sets or COUNTER_SETS):
            d = os.path.join(base, "set%d" % si)
            cmd = [exe, "--pmc"] + cset + ["--kernel-trace", "--output-format", "csv", "-d", d, "--",
                                            sys.executable, os.path.abspath(__file__), "--counter-child", ",".join(workloads),
                                            "--pairs", str(args.pairs), "--k", str(args.k), "--bf-log2", str(args.bf_log2),
                                            "--genes", str(args.genes), "--gene-len", str(args.gene_len), "--on-target", str(args.on_target)]
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
            env["TMPDIR"] = "/tmp"
            # (its own session: on a timeout the whole group goes -- rocprofv3 AND the python child under it, which would otherwise
            #  keep running on the GPU with its indices resident while the timed legs are measured)
            pp = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd="/tmp", start_new_session=True)
            try:
                so, se = pp.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(pp.pid, signal.SIGKILL)      # (pid == pgid: the child leads its session)
                except OSError:
                    pass
                pp.communicate()
                notes.append("set %d timed out" % si)
                continue
            pr = subprocess.CompletedProcess(cmd, pp.returncode, so, se)
            order, mix, cal = None, None, None
            for ln in pr.stdout.splitlines():
                if ln.startswith('{"counter_child"'):
                    order = json.loads(ln)["counter_child"]
                    mix = json.loads(ln).get("valu_mix")
                    cal = json.loads(ln).get("lookup_calibration")
            if pr.returncode != 0 or order is None:
                notes.append("set %d failed (rc %d): %s" % (si, pr.returncode, (pr.stderr or "")[-300:].replace("\n", " | ")))
                continue
            got = parse_counter_dir(d, order)
            if mix and "SQ_INSTS_VALU" in cset:
                # the issue-ceiling kernel's dispatches, in order: (warm-up, timed) per occupancy; VALU instructions per wave-iteration
                vals = parse_kernel_counter(d, "valu_mix_kernel", "SQ_INSTS_VALU")
                if len(vals) == 2 * len(mix):
                    res.setdefault("valu_mix", {})
                    for i, m in enumerate(mix):
                        res["valu_mix"][str(m["waves_per_simd"])] = vals[2 * i + 1] / float(m["wave_iterations_timed_launch"])
            if cal:
                # random_lookup_kernel's dispatches in this pass: (warm-up, timed) x (one read per line, both halves)
                lc = res.setdefault("lookup_calibration", dict(cal))
                for cn in cset:
                    vals = parse_kernel_counter(d, "random_lookup_kernel", cn)
                    if len(vals) == 4:
                        lc[cn + "_per_line_one_read"] = vals[1] / cal["lines"]
                        lc[cn + "_per_line_both_halves"] = vals[3] / cal["lines"]
            for o in order:
                w = o["workload"]
                if w in got:
                    res[w].setdefault("kernel", got[w]["kernel"])
                    for name, part in got[w].get("kernels", {}).items():
                        res[w].setdefault("kernels", {}).setdefault(name, {}).update(part)
                    res[w].setdefault("kernel_reported_by_library", o["kernel"])
                    res[w].setdefault("n_assoc", o["n_assoc"])
                    for cn, v in got[w]["counters"].items():
                        res[w].setdefault(cn, v)
            if not keep_dir:
                shutil.rmtree(d, ignore_errors=True)
    finally:
        if not keep_dir:
            shutil.rmtree(base, ignore_errors=True)
    res = {w: e for w, e in res.items() if len(e) > 1 or w in ("valu_mix", "lookup_calibration")}
    note = "live: rocprofv3 --pmc child passes of this run (%d sets, %.0f s)" % (len(sets or COUNTER_SETS), time.time() - t0)
    if notes:
        note += "; " + "; ".join(notes)
    return res, note


def committed_counters(args):
    """profiles/pmc_counters.json, only when it was taken on the kernel sources being run -> (dict workload -> counters, note)"""
    pfile = os.path.join(ROOT, "profiles", "pmc_counters.json")
    sha = kernel_src_sha()
    if not os.path.exists(pfile):
        return {}, "profiles/pmc_counters.json absent"
    try:
        pj = json.load(open(pfile))
    except Exception as ex:   # a broken profile file must not break the bench line
        return {}, "profiles/pmc_counters.json unreadable: %r" % (ex,)
    if pj.get("kernel_src_sha") != sha:
        return {}, "profiles/pmc_counters.json was taken on other kernel sources (%s, now %s): not used" % (pj.get("kernel_src_sha"), sha)
    out = {}
    for name, e in pj.get("workloads", {}).items():
        if e.get("pairs") != args.pairs:
            continue
        if name == "configs1_ot%.2f" % args.on_target and pj.get("k") == args.k and pj.get("bf_log2") == args.bf_log2:
            out["configs1"] = e
        elif name in WORKLOADS and name != "configs1":
            out[name] = e
    return out, "profiles/pmc_counters.json (commit %s: a separate rocprofv3 --pmc run on the same kernel sources, not this run)" % pj.get("commit")


def fetch_write_bytes(e):
    """FETCH_SIZE / WRITE_SIZE are reported in KB"""
    f = e.get("FETCH_SIZE", e.get("FETCH_SIZE_KB"))
    w = e.get("WRITE_SIZE", e.get("WRITE_SIZE_KB"))
    return (f * 1024.0 if f is not None else None), (w * 1024.0 if w is not None else None)


def instruction_part(e, n_pairs, kern_ms):
    iv = e.get("SQ_INSTS_VALU")
    if not iv:
        return None
    t = kern_ms * 1e-3
    d = {"valu_per_pair": round(iv / n_pairs, 1), "salu_per_pair": round(e.get("SQ_INSTS_SALU", 0) / n_pairs, 1),
         "lds_per_pair": round(e.get("SQ_INSTS_LDS", 0) / n_pairs, 1),
         "G_valu_wave_instructions_per_s": round(iv / t / 1e9, 1), "frac_of_valu_issue_peak": round(iv / t / 1e9 / VALU_PEAK_GINST, 4),
         "simd_cycles_per_valu_instruction": round(t * CLK_GHZ * 1e9 * N_SIMD / iv, 2)}
    ga, gns = e.get("GRBM_GUI_ACTIVE"), e.get("GRBM_PASS_NS")
    if ga and gns:
        # the clock the chip held during this kernel (it lowers it under load): busy cycles of the 8 XCDs / 8 / duration, same pass
        clk = ga / 8.0 / gns
        d["shader_clock_GHz"] = round(clk, 3)
        d["simd_cycles_per_valu_instruction_at_that_clock"] = round(t * clk * 1e9 * N_SIMD / iv, 2)
        d["frac_of_valu_issue_peak_at_that_clock"] = round(iv / t / 1e9 / (N_SIMD * clk / 2.0), 4)
    wc = e.get("SQ_WAVE_CYCLES")
    if wc:
        for key, nm in (("SQ_ACTIVE_INST_ANY", "issuing"), ("SQ_WAIT_INST_ANY", "wait_inst_any"), ("SQ_WAIT_ANY", "wait_any")):
            if e.get(key) is not None:
                d["wave_cycles_" + nm] = round(e[key] / wc, 3)
    return d


def fetch_calibration(lc):
    """the counter child's calibration lookups -> what a random 16-byte lookup behind the caches moves and what FETCH_SIZE says it
    moves.  A KNOWN number of lines is read once with one 16-byte read per 128-byte line and once with a read in each 64-byte half:
    if the second read is free (same rate, same FETCH_SIZE per line) a memory-side request brings the whole line, else half of one.
    -> dict(bytes_per_request, fetch_size_factor, ...) or None when the pass did not deliver the counters"""
    if not lc:
        return None
    f1, f2 = lc.get("FETCH_SIZE_per_line_one_read"), lc.get("FETCH_SIZE_per_line_both_halves")
    g1, g2 = lc.get("G_lines_per_s_one_read"), lc.get("G_lines_per_s_both_halves")
    if not f1 or not f2 or not g1 or not g2:
        return None
    f1b, f2b = f1 * 1024.0, f2 * 1024.0          # (FETCH_SIZE is reported in KB)
    whole_line = g2 >= 0.8 * g1 and f2b <= 1.3 * f1b
    bpr = 128 if whole_line else 64
    out = {"table_bytes": lc.get("table_bytes"), "lines": lc.get("lines"),
           "G_lines_per_s": {"one_read_per_line": g1, "a_read_in_each_half": g2},
           "FETCH_SIZE_bytes_per_line": {"one_read_per_line": round(f1b, 2), "a_read_in_each_half": round(f2b, 2)},
           "TCC_MISS_per_line": {"one_read_per_line": round(lc.get("TCC_MISS_sum_per_line_one_read", 0.0), 3),
                                 "a_read_in_each_half": round(lc.get("TCC_MISS_sum_per_line_both_halves", 0.0), 3)},
           "second_half_is_free": bool(whole_line), "bytes_per_request": bpr, "fetch_size_factor": round(bpr / f1b, 3),
           "how": "shk_measure_random_lookups in the FETCH_SIZE / TCC counter passes of this run: %d random lines of an %d GiB table, 16 bytes read of each, "
                  "then 16 bytes of each 64-byte half; a second half that costs neither time nor FETCH_SIZE means a request brings the 128-byte line, "
                  "else a request is 64 bytes; fetch_size_factor = that / FETCH_SIZE per line" % (lc.get("lines", 0), (lc.get("table_bytes") or 0) >> 30)}
    return out


def request_rate_roofline(e, kern_ms, ceiling, alg_bytes, in_bytes, out_bytes, n_pairs, w, cal=None):
    """an index far beyond the caches: the classify kernel is bound by the RATE of random memory-side requests (one 128-byte line
    per 16-byte bucket).  achieved = L2 misses per second (counter pass), peak = shk_measure_random_lookups in this run."""
    t = kern_ms * 1e-3
    rl = {"bound": "memory-side-request-rate", "achieved": None, "peak": round(ceiling, 2) if ceiling else None, "unit": "G requests/s (random 128-B lines behind L2)",
          "frac": None, "traffic": None, "kernel_ms": round(kern_ms, 4)}
    req = e.get("TCC_MISS_sum") if e else None
    if req and ceiling:
        a = req / t / 1e9
        rl["achieved"] = round(a, 2)
        rl["frac"] = round(min(1.0, a / ceiling), 4)
        if a > ceiling:
            rl["note"] = "measured request rate %.1f G/s exceeds the ceiling measured in this run (%.1f G/s): frac capped at 1" % (a, ceiling)
        rl["memory_side_requests_per_launch"] = int(req)
        rl["requests_per_pair"] = round(req / n_pairs, 1)
        bpr = cal["bytes_per_request"] if cal else L2_LINE_BYTES
        rl["bytes_per_request"] = bpr
        rl["line_traffic_GBps"] = round(bpr * req / t / 1e9, 1)
        rl["line_traffic_frac_of_hbm"] = round(bpr * req / t / 1e9 / HBM_PEAK_GBPS, 4)
    if e:
        fb, wb = fetch_write_bytes(e)
        if fb is not None:
            # FETCH_SIZE's factor for THIS access pattern -- random 16-byte lookups -- calibrated in this run on a known number of them
            # (fetch_calibration); without the calibration the guide's factor for wide streaming reads (2) and a note
            fac = cal["fetch_size_factor"] if cal else 2.0
            traffic = fac * fb + (wb or 0.0)
            rl["traffic"] = int(traffic)
            rl["traffic_how"] = ("%.3g x FETCH_SIZE + WRITE_SIZE per launch; the factor calibrated on random 16-byte lookups in this run (roofline.fetch_calibration)" % fac
                                 if cal else "2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md's factor for wide streaming reads: UNCALIBRATED for this pattern), per launch")
            rl["traffic_over_algorithmic"] = round(traffic / alg_bytes, 2)
            rl["traffic_GBps"] = round(traffic / t / 1e9, 1)
        rl["instructions"] = instruction_part(e, n_pairs, kern_ms)
        rl["kernel"] = e.get("kernel")
        if e.get("kernels") and len(e["kernels"]) > 1:
            # the launch's work is split over two kernels (anchor_verdict_kernel in front of the table kernel): the counters above are their sums
            rl["kernels"] = {name: {cn: (round(v / n_pairs, 1) if cn != "FETCH_SIZE" else round(v * 1024.0)) for cn, v in part.items()} for name, part in e["kernels"].items()}
            rl["kernels_what"] = "per kernel of the launch: VALU / SALU instructions and memory-side requests per pair, FETCH_SIZE in bytes per launch"
        if cal:
            rl["fetch_calibration"] = cal
    rl["hbm_compulsory"] = {"bytes": int(in_bytes + out_bytes), "in": int(in_bytes), "out": int(out_bytes), "GBps": round((in_bytes + out_bytes) / t / 1e9, 1),
                            "frac_of_hbm_peak": round((in_bytes + out_bytes) / t / 1e9 / HBM_PEAK_GBPS, 4),
                            "what": "input bases (+ qualities) + result bytes of one launch / kernel time / 8 TB/s: what any implementation has to move"}
    rl["model_8d"] = {"bytes": int(alg_bytes), "GBps": round(alg_bytes / t / 1e9, 1), "x_of_hbm_peak": round(alg_bytes / t / 1e9 / HBM_PEAK_GBPS, 4),
                      "kmers": int(w["n_kmers"]), "hits": int(w["n_hits"]),
                      "why": "SURVEY 8(d): L(1+[Q]) + 8 per k-mer + (16 + 2 l) per hit + 8 per read over EVERY k-mer of every read, as the reference visits "
                             "them; the bound cut, the early decision, the anchored extension and the partial round make fewer lookups than that"}
    return rl


# =====================================================================================================================
# the command itself, end to end
# =====================================================================================================================
def mem_budget_bytes(base):
    """what may be put into `base` (a tmpfs: it is memory): the smallest of the file system's free space, the cgroup's headroom and
    MemAvailable, and of that at most 45 %"""
    import shutil
    lim = [shutil.disk_usage(base).free]
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                lim.append(int(ln.split()[1]) * 1024)
    except OSError:
        pass
    for mx, cur in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                    ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            m = open(mx).read().strip()
            if m != "max":
                lim.append(int(m) - int(open(cur).read().strip()))
        except (OSError, ValueError):
            pass
    return int(0.45 * min(lim))


def render_ssv(ridx, gids, nd):
    """"r<nd digits>/1 gene<id>\n" per association (ReadOutput.hpp:43: mate 1's id, the gene's name), ids of any number of digits:
    a fixed-width byte matrix with NUL where a leading digit is not printed, flattened without the NULs"""
    import numpy as np
    gi = gids.astype(np.int64)
    txt = np.zeros((len(ridx), 1 + nd + 3 + 4 + 5 + 1), np.uint8)
    txt[:, 0] = ord("r")
    for d in range(nd):
        txt[:, 1 + d] = ord("0") + (ridx // 10 ** (nd - 1 - d)) % 10
    txt[:, 1 + nd:4 + nd] = np.frombuffer(b"/1 ", np.uint8)
    txt[:, 4 + nd:8 + nd] = np.frombuffer(b"gene", np.uint8)
    for d in range(5):
        p10 = 10 ** (4 - d)
        dig = (gi // p10) % 10
        shown = (gi >= p10) | (d == 4)
        txt[:, 8 + nd + d] = np.where(shown, ord("0") + dig, 0)
    txt[:, 13 + nd] = 10
    flat = txt.ravel()
    return flat[flat != 0].tobytes()


def cli_end_to_end(args, genes, dev, h, L, n_gpus=1, ots=None, only_small=False, with_gz=True, with_shared=True, extra_args=(), alternate=0, k=None):
    """(see below)  ots: the on-target rates to run (default 0.02 and 0.50); only_small: one sample size; extra_args: more options of the
    command (-b 8 for the configs[2] index); alternate: pairs per stretch of a sample whose stretches are drawn from the genes and from
    elsewhere by turns (the on-target rate reads "alternating")."""
    return _cli_end_to_end(args, genes, dev, h, L, n_gpus, ots, only_small, with_gz, with_shared, list(extra_args), alternate, k if k is not None else args.k)


def _cli_end_to_end(args, genes, dev, h, L, n_gpus, ots, only_small, with_gz, with_shared, extra_args, alternate, k_opt):
    """`shark` (shark_amd/bin/shark, the reference's command line: main.cpp:83-240, README.md:47-52) on FASTQ files, run as a user runs
    it, wall time of the whole process taken from outside.  Two on-target rates (0.02: one gene against a whole sample; 0.50: the
    north-star read mix, where ReadOutput.hpp:37-50 writes half the sample out again) x two sample sizes (so the fixed costs --
    HIP initialisation, process exit -- and the steady state separate).  The ssv must equal, byte for byte (md5), the text rendered
    from the associations the library returns for the same pairs resident in HBM; the output FASTQ must have exactly the bytes of
    the associated records.  n_gpus > 1: the host-fed multi-GPU leg (`shark --gpus N`, one host feeding N GPUs, ordered drain)."""
    import shutil
    import subprocess
    import tempfile
    import numpy as np
    import torch
    from shark_amd import synth
    from shark_amd.capi import hip_memcpy_dtoh
    exe = os.path.join(ROOT, "shark_amd", "bin", "shark")
    base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    nd = 9
    H = 2 + nd + 3
    W = H + L + 3 + L + 1
    small = args.cli_pairs
    chunk = small                       # the samples are generated chunk by chunk; the large one is a multiple of the small one
    budget = mem_budget_bytes(base)
    # the large sample: input 2 n W, output FASTQ at most on_target x that, ssv 20 B per line
    big = args.cli_big_pairs
    while big > small and 2 * big * W * 1.55 > budget:
        big -= chunk
    sizes = [small] + ([big] if big > small and not only_small else [])
    if 2 * small * W * 1.55 > budget:
        return {"skipped": "not enough room in %s for %d pairs (budget %d bytes)" % (base, small, budget)}
    same_len_names = True          # (the ssv is always rendered and compared: render_ssv)
    gen_chunk = alternate if alternate else chunk
    threads_list = [min(16 * n_gpus, host_threads())] if n_gpus > 1 else [min(12, host_threads())]
    td = tempfile.mkdtemp(dir=base)
    out = {"what": "shark_amd/bin/shark -r g.fa -1 r1.fq -2 r2.fq -o o1.fq -p o2.fq -t T%s > out.ssv on synthetic pairs 2x%d bp in %s (files written and read once "
                   "beforehand, untimed); wall time of the whole process from outside; ssv md5 = md5 of the text rendered from the device-resident result"
                   % (" --gpus %d" % n_gpus if n_gpus > 1 else "", L, base),
           "n_gpus": n_gpus, "record_bytes": W, "tmpfs_budget_bytes": budget, "runs": []}
    try:
        with open(os.path.join(td, "g.fa"), "wb") as f:
            for gi, g in enumerate(genes):
                f.write(b">gene%d\n" % gi + g.tobytes() + b"\n")
        for ot in (ots if ots is not None else ([0.02] if n_gpus > 1 else [0.02, 0.50])):
            t0 = time.time()
            n_big = sizes[-1]
            ot_label = "alternating" if alternate else ot
            # the compressed leg (ordinary gzip -1 files, one member each: inflated in parallel in two passes, gzip_parallel.hpp) on the
            # first gz_n pairs of the 0.02 sample
            gz_n = min(args.cli_gz_pairs, small) if (with_gz and ot == 0.02 and n_gpus == 1 and shutil.which("gzip")) else 0
            gz_md5, gz_lines, gz_assoc_reads = None, 0, 0
            log("  cli: generating %d pairs at on-target %s" % (n_big, ot_label))
            md5 = hashlib.md5()
            md5_at, lines_at, assoc_reads_at = {}, {}, {}
            lines = assoc_reads = 0
            f1 = open(os.path.join(td, "r1.fq"), "wb")
            f2 = open(os.path.join(td, "r2.fq"), "wb")
            for c0 in range(0, n_big, gen_chunk):
                m_all = min(gen_chunk, n_big - c0)
                ot_c = (1.0 if (c0 // gen_chunk) % 2 == 0 else 0.0) if alternate else ot      # (alternating: a stretch from the genes, a stretch from elsewhere)
                b = synth.make_pairs_device(m_all, genes, dev, seed=synth.SEED + 99 + c0 // gen_chunk + int(ot * 1000), read_len=L, on_target=ot_c)
                torch.cuda.synchronize()
                r = h.classify_device(m_all, b["seq1"].data_ptr(), b["off1"].data_ptr(), b["seq2"].data_ptr(), b["off2"].data_ptr(), max_read_len=L)
                goff = np.empty(m_all + 1, np.uint32)
                hip_memcpy_dtoh(goff, r.gene_off, goff.nbytes)
                gids = np.empty(max(int(r.n_assoc), 1), np.uint16)
                hip_memcpy_dtoh(gids, r.gene_ids, int(r.n_assoc) * 2)
                gids = gids[:int(r.n_assoc)]
                cnt = np.diff(goff.astype(np.int64))
                lines += int(r.n_assoc)
                assoc_reads += int((cnt > 0).sum())
                if c0 == 0 and gz_n:    # (the compressed leg classifies the first gz_n pairs of this sample)
                    gz_lines = int(goff[gz_n])
                    gz_assoc_reads = int((cnt[:gz_n] > 0).sum())
                if same_len_names:      # "r<9 digits>/1 gene<id>\n" per association, reads in input order, genes ascending (ReadOutput.hpp:43)
                    for a0 in range(0, m_all, 4_000_000):       # (rendered 4 M pairs at a time)
                        a1 = min(m_all, a0 + 4_000_000)
                        ridx = np.repeat(np.arange(c0 + a0, c0 + a1, dtype=np.int64), cnt[a0:a1])
                        md5.update(render_ssv(ridx, gids[int(goff[a0]):int(goff[a1])], nd))
                    if c0 == 0 and gz_n:
                        gz_md5 = hashlib.md5(render_ssv(np.repeat(np.arange(0, gz_n, dtype=np.int64), cnt[:gz_n]), gids[:int(goff[gz_n])], nd)).hexdigest()
                idx = torch.arange(c0, c0 + m_all, device=dev, dtype=torch.int64)
                for mate, key, fh in ((1, "seq1", f1), (2, "seq2", f2)):
                    for s0 in range(0, m_all, 2_000_000):            # "@r<9 digits>/<mate>\n" + bases + "\n+\n" + qualities + "\n", 2 M records at a time
                        m = min(2_000_000, m_all - s0)
                        rec = torch.empty((m, W), dtype=torch.uint8, device=dev)
                        rec[:, 0] = ord("@")
                        rec[:, 1] = ord("r")
                        ii = idx[s0:s0 + m]
                        for d in range(nd):
                            rec[:, 2 + d] = (ord("0") + (ii // 10 ** (nd - 1 - d)) % 10).to(torch.uint8)
                        rec[:, H - 3] = ord("/")
                        rec[:, H - 2] = ord("0") + mate
                        rec[:, H - 1] = 10
                        rec[:, H:H + L] = b[key][s0 * L:(s0 + m) * L].view(m, L)
                        rec[:, H + L] = 10
                        rec[:, H + L + 1] = ord("+")
                        rec[:, H + L + 2] = 10
                        rec[:, H + L + 3:H + 2 * L + 3] = ord("I")
                        rec[:, H + 2 * L + 3] = 10
                        rec.cpu().numpy().tofile(fh)
                        del rec
                del b
                done = c0 + m_all
                if done in sizes:
                    md5_at[done], lines_at[done], assoc_reads_at[done] = md5.hexdigest(), lines, assoc_reads
            f1.close()
            f2.close()
            for mate in (1, 2):   # one untimed read: the first read of freshly written tmpfs pages pays for their LRU activation, four times a later one
                subprocess.run(["cat", os.path.join(td, "r%d.fq" % mate)], stdout=subprocess.DEVNULL)
            gen_s = time.time() - t0
            for n in reversed(sizes):            # the large sample first; the small one is its prefix (the files are cut, nothing is copied)
                if n != n_big:
                    for mate in (1, 2):
                        os.truncate(os.path.join(td, "r%d.fq" % mate), n * W)
                # workers: the command's own N (one per GPU), or -- on one GPU, small sample at 0.02 -- also two workers SHARING the
                # device (`--devices 0,0`): the N-worker code paths (queues, ordered drain, parallel finalize) in front of the driver
                worker_sets = [args.cli_devices]
                if with_shared and n_gpus == 1 and ot == 0.02 and n == small and not args.cli_devices:
                    worker_sets.append("0,0")
                for threads, devices in [(t, d) for t in threads_list for d in worker_sets]:
                    for fn in ("o1.fq", "o2.fq", "out.ssv"):
                        try:
                            os.unlink(os.path.join(td, fn))
                        except OSError:
                            pass
                    cmd = [exe, "-r", os.path.join(td, "g.fa"), "-1", os.path.join(td, "r1.fq"), "-2", os.path.join(td, "r2.fq"),
                           "-o", os.path.join(td, "o1.fq"), "-p", os.path.join(td, "o2.fq"), "-k", str(k_opt), "-v", "-t", str(threads)] + extra_args
                    if devices:
                        cmd += ["--devices", devices]
                    elif n_gpus > 1:
                        cmd += ["--gpus", str(n_gpus)]
                    log("  cli: running on %d pairs%s" % (n, " (--devices %s)" % devices if devices else ""))
                    t0 = time.time()
                    with open(os.path.join(td, "out.ssv"), "wb") as so:
                        pr = subprocess.run(cmd, stdout=so, stderr=subprocess.PIPE)
                    dt = time.time() - t0
                    got_md5, got_lines = hashlib.md5(), 0
                    with open(os.path.join(td, "out.ssv"), "rb") as so:
                        for blk in iter(lambda: so.read(1 << 24), b""):
                            got_md5.update(blk)
                            got_lines += blk.count(b"\n")
                    stages, busy = {}, None
                    for ln in pr.stderr.decode(errors="replace").splitlines():
                        if ln.startswith("[shark/ms] "):
                            nm, ms = ln[11:].split(" (epoch")[0].rsplit(" ", 1)
                            stages[nm] = round(float(ms) / 1e3, 3)
                        elif ln.startswith("[shark/gpu-busy]"):
                            busy = [round(float(x), 3) for x in ln.split()[1:]]
                    fq_bytes = sum(os.path.getsize(os.path.join(td, fn)) if os.path.exists(os.path.join(td, fn)) else 0 for fn in ("o1.fq", "o2.fq"))
                    ok_md5 = (got_md5.hexdigest() == md5_at[n]) if same_len_names else None
                    ok = pr.returncode == 0 and got_lines == lines_at[n] and ok_md5 is not False and fq_bytes == 2 * assoc_reads_at[n] * W
                    run = {"pairs": n, "on_target": ot_label, "threads": threads, "devices": devices, "wall_s": round(dt, 3), "value": round(2 * n / dt, 1), "unit": "reads/s", "rc": pr.returncode,
                           "ssv_lines": got_lines, "expected_ssv_lines": lines_at[n], "ssv_md5": got_md5.hexdigest(), "ssv_md5_equals_device_result": ok_md5,
                           "fastq_out_bytes": fq_bytes, "fastq_out_bytes_expected": 2 * assoc_reads_at[n] * W, "valid": ok,
                           "stage_s_since_start": stages, "input_bytes": 2 * n * W, "generate_s": round(gen_s, 1)}
                    if "contexts created" in stages and "contexts destroyed" in stages:
                        # fixed costs: HIP initialisation up to "contexts created" (the readers already parse meanwhile) and what is left of the
                        # process behind "contexts destroyed" (exit)
                        st = stages["contexts destroyed"] - stages["contexts created"]
                        run["fixed_s"] = {"until_contexts_created": stages["contexts created"], "exit": round(max(0.0, dt - stages["contexts destroyed"]), 3)}
                        run["steady_reads_per_s"] = round(2 * n / st, 1) if st > 0 else None
                    if busy is not None:
                        run["gpu_busy_s"] = busy          # per GPU: seconds its analyzer thread spent in shk_classify_submit / _wait
                    out["runs"].append(run)
            if gz_n:
                log("  cli: gzip -1 of the first %d pairs" % gz_n)
                t0 = time.time()
                ps = []
                for mate in (1, 2):
                    with open(os.path.join(td, "r%d.fq" % mate), "rb") as fi, open(os.path.join(td, "g%d.fq.gz" % mate), "wb") as fo:
                        hd = subprocess.Popen(["head", "-c", str(gz_n * W)], stdin=fi, stdout=subprocess.PIPE)
                        ps.append((hd, subprocess.Popen(["gzip", "-1"], stdin=hd.stdout, stdout=fo)))
                        hd.stdout.close()
                ok_gz = all(g.wait() == 0 for _, g in ps)
                gz_s = time.time() - t0
                gz_bytes = sum(os.path.getsize(os.path.join(td, "g%d.fq.gz" % mate)) for mate in (1, 2))
                for mate in (1, 2):
                    subprocess.run(["cat", os.path.join(td, "g%d.fq.gz" % mate)], stdout=subprocess.DEVNULL)
                threads = 16
                for fn in ("o1.fq", "o2.fq", "out.ssv"):
                    try:
                        os.unlink(os.path.join(td, fn))
                    except OSError:
                        pass
                cmd = [exe, "-r", os.path.join(td, "g.fa"), "-1", os.path.join(td, "g1.fq.gz"), "-2", os.path.join(td, "g2.fq.gz"),
                       "-o", os.path.join(td, "o1.fq"), "-p", os.path.join(td, "o2.fq"), "-k", str(args.k), "-v", "-t", str(threads)]
                log("  cli: running on the compressed files")
                t0 = time.time()
                with open(os.path.join(td, "out.ssv"), "wb") as so:
                    pr = subprocess.run(cmd, stdout=so, stderr=subprocess.PIPE)
                dt = time.time() - t0
                got = open(os.path.join(td, "out.ssv"), "rb").read()
                fq_bytes = sum(os.path.getsize(os.path.join(td, fn)) if os.path.exists(os.path.join(td, fn)) else 0 for fn in ("o1.fq", "o2.fq"))
                ok_md5 = (hashlib.md5(got).hexdigest() == gz_md5) if same_len_names else None
                out["gzip_input"] = {"what": "the same command on `gzip -1` files (one gzip member per file) of the first %d pairs of the 0.02 sample, -t %d" % (gz_n, threads),
                                     "pairs": gz_n, "threads": threads, "compressed_bytes": gz_bytes, "text_bytes": 2 * gz_n * W, "gzip_s": round(gz_s, 1),
                                     "wall_s": round(dt, 3), "value": round(2 * gz_n / dt, 1), "unit": "reads/s", "rc": pr.returncode,
                                     "ssv_lines": got.count(b"\n"), "expected_ssv_lines": gz_lines, "ssv_md5_equals_device_result": ok_md5,
                                     "valid": bool(ok_gz and pr.returncode == 0 and got.count(b"\n") == gz_lines and ok_md5 is not False
                                                   and fq_bytes == 2 * gz_assoc_reads * W)}
                for fn in ("g1.fq.gz", "g2.fq.gz"):
                    os.unlink(os.path.join(td, fn))
            for fn in ("r1.fq", "r2.fq", "o1.fq", "o2.fq", "out.ssv"):
                try:
                    os.unlink(os.path.join(td, fn))
                except OSError:
                    pass
        valid = [x for x in out["runs"] if x["valid"]]
        if "gzip_input" in out and not out["gzip_input"]["valid"]:
            valid = []          # (an invalid compressed run invalidates the leg like any other)
        out["all_runs_valid"] = len(valid) == len(out["runs"]) and bool(valid)
        # two sample sizes: the slope is the steady state, the intercept the fixed cost
        by = {}
        for x in valid:
            if x["devices"] == args.cli_devices:
                by.setdefault((x["on_target"], x["threads"]), []).append(x)
        slopes = []
        for (ot, th), xs in sorted(by.items()):
            xs.sort(key=lambda x: x["pairs"])
            if len(xs) >= 2 and xs[-1]["wall_s"] > xs[0]["wall_s"]:
                sl = 2 * (xs[-1]["pairs"] - xs[0]["pairs"]) / (xs[-1]["wall_s"] - xs[0]["wall_s"])
                slopes.append({"on_target": ot, "threads": th, "steady_reads_per_s_from_two_sizes": round(sl, 1),
                               "fixed_s_from_two_sizes": round(xs[0]["wall_s"] - 2 * xs[0]["pairs"] / sl, 3)})
        out["two_size_fit"] = slopes
        # headline of this leg: the run a user's sample looks like least flattering -- the SMALL sample at 0.02 (fixed costs included)
        pick = [x for x in valid if x["on_target"] == 0.02 and x["pairs"] == small and x["devices"] == args.cli_devices]
        if not pick and ots is not None:
            pick = [x for x in valid if x["pairs"] == small and x["devices"] == args.cli_devices]      # (a leg with rates of its own)
        if pick:
            best = max(pick, key=lambda x: x["value"])
            out.update({k: best[k] for k in ("pairs", "on_target", "threads", "wall_s", "value", "unit")})
        else:
            out.update({"value": None, "unit": "reads/s", "error": "no valid run: see runs[].rc / ssv_md5_equals_device_result / fastq_out_bytes"})
        shared = [x for x in out["runs"] if x["devices"] and x["devices"] != args.cli_devices]
        if shared:
            out["two_workers_on_one_device"] = {"value": shared[0]["value"], "unit": "reads/s", "valid": shared[0]["valid"], "gpu_busy_s": shared[0].get("gpu_busy_s"),
                                                "what": "--devices 0,0: two workers (contexts, pipelines) sharing the GPU; ssv md5 / FASTQ bytes checked like every run"}
        half = [x for x in valid if x["on_target"] == 0.50]
        if half:
            out["value_at_on_target_0.50"] = max(x["value"] for x in half)
            out["steady_at_on_target_0.50"] = max((x.get("steady_reads_per_s") or 0.0) for x in half)
        return out
    finally:
        shutil.rmtree(td, ignore_errors=True)


def position_table_bytes(n_set):
    """the size of an index's position table (index_build.hip): the smallest power of two of 16-byte buckets, two slots each, that
    holds n_set keys at a load of at most 0.3"""
    lg = 10
    while n_set > 0.3 * 2.0 * (1 << lg):
        lg += 1
    return 16 << lg


def boundary_leg(h, batch, nb, L, kinds=("pageable", "pinned"), n_batches=12):
    """PCIe-inclusive rate of the host-buffer entry points on context h (never `value`): shk_classify_submit / shk_classify_wait over
    the first nb pairs of `batch` copied to host buffers, two sets used alternately, SHK_PIPE_DEPTH batches in flight"""
    import numpy as np
    import torch
    from shark_amd import synth
    from shark_amd.capi import SHK_PIPE_DEPTH
    hbp = synth.to_host_sample(batch, nb, L)
    boundary = {"pairs_per_batch": nb, "batches": n_batches, "in_flight": SHK_PIPE_DEPTH,
                "what": "shk_classify_submit / shk_classify_wait over host buffers: H2D + kernels + D2H, results on the host; "
                        "the H2D of the next batches overlaps the kernels of the current one"}
    for kind in kinds:
        bufs = []
        for rep in range(2):                      # two sets of host buffers, used alternately
            arrs = {}
            for kk in ("seq1", "seq2"):
                t = torch.from_numpy(hbp[kk]).clone()
                arrs[kk] = (t.pin_memory() if kind == "pinned" else t).numpy()
            for kk in ("off1", "off2"):
                t = torch.from_numpy(hbp[kk].view(np.int64)).clone()
                arrs[kk] = (t.pin_memory() if kind == "pinned" else t).numpy().view(np.uint64)
            bufs.append(arrs)

        def stream(nbatches):
            tickets, assoc = [], 0
            for i in range(nbatches):
                if len(tickets) == SHK_PIPE_DEPTH:
                    assoc += int(h.wait(tickets.pop(0), copy=False)[0][-1])
                a = bufs[i % 2]
                tickets.append(h.submit(a["seq1"], a["off1"], a["seq2"], a["off2"]))
            while tickets:
                assoc += int(h.wait(tickets.pop(0), copy=False)[0][-1])
            return assoc
        stream(3)
        t0 = time.perf_counter()
        assoc = stream(n_batches)
        tb = (time.perf_counter() - t0) / n_batches
        boundary[kind] = {"value": round(2 * nb / tb, 1), "unit": "reads/s", "ms_per_batch": round(tb * 1e3, 2),
                          "GBps_h2d": round((hbp["seq1"].nbytes + hbp["seq2"].nbytes) / tb / 1e9, 1), "assoc_per_batch": assoc // n_batches}
    return boundary


# =====================================================================================================================
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--total-pairs", type=int, default=80_000_000, help="strong scaling: pairs of the read set over ALL GPUs")
    ap.add_argument("--reps-per-step", type=int, default=16, help="passes over the read set per step (the same at every N)")
    ap.add_argument("--pairs", type=int, default=LAUNCH_PAIRS, help="pairs per launch (and per GPU per pass with --scaling weak)")
    ap.add_argument("--k", type=int, default=17)
    ap.add_argument("--bf-log2", type=int, default=33)
    ap.add_argument("--genes", type=int, default=1)
    ap.add_argument("--gene-len", type=int, default=20000)
    ap.add_argument("--on-target", type=float, default=0.5, help="fraction of pairs drawn from a gene")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the configs[2] / configs[4] workloads")
    ap.add_argument("--no-boundary", action="store_true", help="skip the host-buffer (PCIe-inclusive) measurement")
    ap.add_argument("--no-trimmed", action="store_true", help="skip the trimmed-batch leg (its launches include no-op launches of the headline kernel: "
                                                              "a kernel trace of the run then averages them in)")
    ap.add_argument("--no-cli", action="store_true", help="skip the end-to-end run of the shark command line")
    ap.add_argument("--no-async", action="store_true", help="skip the pipelined device-resident leg (value_async)")
    ap.add_argument("--no-live-counters", action="store_true", help="do not run the rocprofv3 --pmc child passes (committed counters are used when they match)")
    ap.add_argument("--cli-pairs", type=int, default=16_000_000)
    ap.add_argument("--cli-devices", default=None, help="device list handed to `shark --devices` in the CLI leg (default: the command's own 0..N-1; "
                    "e.g. 0,0,0,0 rehearses four workers on one GPU)")
    ap.add_argument("--cli-gz-pairs", type=int, default=8_000_000, help="pairs of the CLI leg's run on gzip-compressed files (0 = none)")
    ap.add_argument("--cli-big-pairs", type=int, default=64_000_000, help="the larger of the CLI leg's two samples (cut to what /dev/shm may hold)")
    ap.add_argument("--cpu-sample-pairs", type=int, default=0, help="0 = threads x 50 000 (one reference chunk per thread)")
    ap.add_argument("--counter-child", default="", help=argparse.SUPPRESS)
    ap.add_argument("--profile-passes", default="", help="write the counter passes of all workloads to this JSON file and exit (tools/gpu_profiles.sh)")
    args = ap.parse_args()
    if args.counter_child:
        return counter_child(args)
    if args.profile_passes:
        # (tools/gpu_profiles.sh: the same passes bench.py makes live, plus a fourth SQ set, at three on-target rates of the headline)
        res = {}
        for ot in (0.0, 0.5, 1.0):
            args.on_target = ot
            r, note = collect_counters(args, ["configs1"], sets=COUNTER_SETS_FULL)
            res["configs1_ot%.2f" % ot] = r.get("configs1", {})
            print(note, file=sys.stderr, flush=True)
        args.on_target = 0.5
        r, note = collect_counters(args, ["configs2", "configs2_ot1.00", "configs4_uniform", "configs4_ends"], sets=COUNTER_SETS_FULL, timeout_s=600)
        res.update(r)
        print(note, file=sys.stderr, flush=True)
        json.dump(res, open(args.profile_passes, "w"), indent=1)
        return
    if args.gpus < 1:
        ap.error("--gpus must be at least 1")
    if args.reps_per_step < 1:
        ap.error("--reps-per-step must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)           # (does not return)

    import numpy as np
    import torch
    from shark_amd import SharkHip
    from shark_amd import dist as sdist
    from shark_amd import synth
    from shark_amd.capi import hip_memcpy_dtoh

    rank, local_rank, world = sdist.env_rank()
    if world != args.gpus:
        # the line's n_gpus must be what was asked for: a launcher that started another number of ranks is an error, not a warning
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (start it as `python3 bench.py --gpus N`, or with a launcher whose "
                 "--nproc-per-node equals --gpus)" % (args.gpus, world))
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        sys.exit("bench.py: no GPU (there is no CPU path)")
    if world > n_dev and sdist.backend_name() == "nccl":
        sys.exit("bench.py: --gpus %d but this node has %d GPU(s); one process per GPU over RCCL needs a GPU per rank "
                 "(SHARK_DIST_BACKEND=gloo rehearses the N-rank path on fewer)" % (world, n_dev))
    local_rank %= n_dev                    # (a gloo dry run may put several ranks on one GPU)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sdist.init(sdist.backend_name(), dev)

    k, c, bf_bits = args.k, 0.6, 1 << args.bf_log2
    L = 150
    lp = args.pairs
    reps = args.reps_per_step

    # ---- which chunks of the read set this rank owns -----------------------------
    chunk_pairs, my_chunks, pairs_per_pass_all = shard_chunks(args.scaling, args.total_pairs, lp, world, rank)

    def barrier():
        torch.cuda.synchronize()
        sdist.barrier()
        torch.cuda.synchronize()

    done_flag = os.path.join("/tmp", "shark_bench_%s.done" % os.environ.get("MASTER_PORT", "0"))
    if rank == 0 and os.path.exists(done_flag):
        os.unlink(done_flag)               # (left by an earlier job on the same port; the other ranks look for it only much later)

    # ---- index: replicated by deterministic rebuild on every GPU (not timed) -------
    genes = synth.make_reference(args.genes, args.gene_len)
    t0 = time.time()
    h = SharkHip(k=k, c=c, bf_bits=bf_bits, device=local_rank)
    info = h.build([g.tobytes() for g in genes])
    t_build = time.time() - t0
    h.dist_init(sdist)                                 # RCCL communicator inside the library (no-op for world 1)
    comm_rank, ranks_seen = h.dist_info()              # what that communicator says (ncclCommCount), not what the environment said
    if ranks_seen != world or comm_rank != rank:
        sys.exit("bench.py: rank %d of %d, but the communicator reports rank %d of %d" % (rank, world, comm_rank, ranks_seen))

    # ---- this rank's shard of the read set, generated in HBM ------------------------
    batches = [synth.make_pairs_device(chunk_pairs, genes, dev, seed=synth.SEED + 1 + cidx, read_len=L, on_target=args.on_target)
               for cidx in my_chunks]
    torch.cuda.synchronize()
    ptrs = [{kk: (v.data_ptr() if v is not None else 0) for kk, v in b.items()} for b in batches]

    def one_pass(hh, pp, n):
        res, a = None, 0
        for p in pp:
            res = hh.classify_device(n, p["seq1"], p["off1"], p["seq2"], p["off2"], p["qual1"], p["qual2"], max_read_len=L)
            a += int(res.n_assoc)
        return res, a

    def timed(hh, pp, n, steps, warmup, n_reps, reduce_counts):
        """-> (max-over-ranks seconds, library timing, last result, local associations, reduced counts, this rank's split in ms)"""
        hh.timing_enable(True)           # (on during the warm-up too: the event pairs the timed steps use exist before the window opens)
        for _ in range(warmup):
            for _ in range(n_reps):
                one_pass(hh, pp, n)
        if reduce_counts:
            hh.dist_gene_counts_allreduce()      # warm-up collective, OUTSIDE the window: RCCL sets its channels up on a communicator's first one
        hh.gene_counts_reset()
        hh.timing_enable(True)
        barrier()
        t0 = time.perf_counter()
        res, n_assoc = None, 0
        for _ in range(steps):
            for _ in range(n_reps):
                res, a = one_pass(hh, pp, n)
                n_assoc += a
        t1 = time.perf_counter()
        counts = hh.dist_gene_counts_allreduce() if reduce_counts else None    # the path's one exchange step (RCCL)
        t2 = time.perf_counter()
        barrier()
        t3 = time.perf_counter()
        dt = t3 - t0
        tm = hh.timing()
        hh.timing_enable(False)
        split = [tm["total_ms"], (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, float(tm["n_launches"])]
        return sdist.max_over_ranks(dt, dev), tm, res, n_assoc, counts, split

    log("index built (%.2f s), %d chunk(s) of %d pairs generated; timing %d warm-up + %d steps of %d passes" % (t_build, len(my_chunks), chunk_pairs, args.warmup, args.steps, reps))
    dt, tm, res, n_assoc_local, counts, split = timed(h, ptrs, chunk_pairs, args.steps, args.warmup, reps, True)
    log("timed window done: %.3f s" % dt)
    n_assoc = torch.tensor([n_assoc_local], dtype=torch.int64, device=dev)
    sdist.allreduce_sum_(n_assoc)
    rows = sdist.gather_rows(split, dev)                 # every rank's split, over the job's own channel
    per_rank = [{"rank": r, "kernel_ms": round(x[0], 3), "wall_ms": round(x[1], 3), "allreduce_ms": round(x[2], 3),
                 "barrier_wait_ms": round(x[3], 3), "launches": int(x[4])} for r, x in enumerate(rows)]
    reads_per_step = 2 * pairs_per_pass_all * reps
    value = reads_per_step * args.steps / dt
    # ---- the same passes through the device-resident PIPELINE (shk_classify_device_submit / shk_classify_wait: three launches in
    # flight, no host synchronisation between two of them, the uniform read length vouched for by the caller as the host entry
    # points find it out themselves).  Never `value`: `value_async` next to it, with what a launch costs either way. ----
    value_async = None
    if world == 1 and not args.no_async:
        from shark_amd.capi import SHK_PIPE_DEPTH
        def async_passes(n_pass):
            tickets, a = [], 0
            for _ in range(n_pass):
                for p in ptrs:
                    if len(tickets) == SHK_PIPE_DEPTH:
                        a += int(h.wait_device(tickets.pop(0)).n_assoc)
                    tickets.append(h.submit_device(chunk_pairs, p["seq1"], p["off1"], p["seq2"], p["off2"], p["qual1"], p["qual2"], max_read_len=L,
                                                   uniform_len1=L, uniform_len2=L))
            while tickets:
                a += int(h.wait_device(tickets.pop(0)).n_assoc)
            return a
        async_passes(max(1, args.warmup) * reps)
        torch.cuda.synchronize()
        ta = time.perf_counter()
        a_async = async_passes(args.steps * reps)
        torch.cuda.synchronize()
        ta = time.perf_counter() - ta
        n_launch = args.steps * reps * len(ptrs)
        value_async = {"value": round(reads_per_step * args.steps / ta, 1), "unit": "reads/s", "ms_per_launch": round(ta * 1e3 / n_launch, 4),
                       "sync_ms_per_launch": round(dt * 1e3 / n_launch, 4), "kernel_ms_per_launch": round(tm["total_ms"] / max(tm["n_launches"], 1), 4),
                       "assoc_equal_to_the_synchronous_passes": a_async == n_assoc_local,
                       "what": "the timed passes again through shk_classify_device_submit / shk_classify_wait, %d launches in flight, read length vouched for "
                               "(no pass over the offsets); wall time of %d launches" % (SHK_PIPE_DEPTH, n_launch)}
        if a_async != n_assoc_local:
            sys.exit("bench.py: the pipelined passes returned %d associations, the synchronous ones %d" % (a_async, n_assoc_local))
    kern_ms = tm["total_ms"] / max(tm["n_launches"], 1)
    headline_kernel = h.last_kernel()

    # ---- trimmed samples on the headline index: the first chunk with a share of its mates cut to 100-150 bases (rank 0, N = 1) -----
    # (reads of any length in one stream, FastqSplitter.hpp:47-93; DESIGN.md 3: sorted by length on the device, classified class by class)
    trimmed = None
    if world == 1 and not args.no_configs and not args.no_trimmed and L == 150 and batches:
        log("trimmed batches")
        trimmed = {"what": "the headline workload's first %d pairs with a share of the mates cut to a random length in [100, 150] (offsets from HBM: the device "
                           "decides what the batch is); ms per launch = the classify launches + the passes over the offsets in front of them "
                           "(shk_timing.prepass_ms), two untimed calls first" % chunk_pairs, "runs": []}
        gq = torch.Generator(device=dev)
        gq.manual_seed(7)
        for share in (0.2, 1.0):
            tb = {}
            for key in ("seq1", "seq2"):
                ln = torch.randint(100, 151, (chunk_pairs,), generator=gq, device=dev)
                ln = torch.where(torch.rand(chunk_pairs, generator=gq, device=dev) < 1.0 - share, torch.full_like(ln, L), ln)
                keep = torch.arange(L, device=dev)[None, :] < ln[:, None]
                tb[key] = batches[0][key].view(chunk_pairs, L)[keep].contiguous()
                off = torch.zeros(chunk_pairs + 1, dtype=torch.int64, device=dev)
                off[1:] = torch.cumsum(ln, 0)
                tb["off" + key[-1]] = off
                del keep, ln
            torch.cuda.synchronize()
            call = lambda: h.classify_device(chunk_pairs, tb["seq1"].data_ptr(), tb["off1"].data_ptr(), tb["seq2"].data_ptr(), tb["off2"].data_ptr(), 0, 0, max_read_len=L)
            call()
            call()          # (the class-by-class path follows the stream: the first ragged batch behind uniform ones takes the ragged instantiation)
            h.timing_enable(True)
            for _ in range(4):
                rt = call()
            tmt = h.timing()
            h.timing_enable(False)
            nl = max(tmt["n_launches"], 1)
            trimmed["runs"].append({"mates_trimmed": share, "ms_per_launch": round((tmt["total_ms"] + tmt["prepass_ms"]) / nl, 3),
                                    "of_which_prepass_ms": round(tmt["prepass_ms"] / nl, 3), "kernel": h.last_kernel(), "n_assoc": int(rt.n_assoc),
                                    "reads_per_s": round(2 * chunk_pairs / ((tmt["total_ms"] + tmt["prepass_ms"]) / nl / 1e3), 1)})
            del tb
        trimmed["untrimmed_ms_per_launch"] = round(kern_ms, 3)
        torch.cuda.empty_cache()

    # ---- BASELINE configs[2] index (60 000 genes, 2^36 bits): every rank, 10 M-pair steps -----
    cfg2 = cfg4 = None
    big = {}
    if not args.no_configs:
        h.close()
        for b in batches[1:]:
            b.clear()
        g2 = synth.make_gencode_like_reference(60000)
        log("configs[2] index")
        t0 = time.time()
        h2 = SharkHip(k=17, c=0.6, bf_bits=1 << 36, device=local_rank)
        info2 = h2.build([g.tobytes() for g in g2])
        t_build2 = time.time() - t0
        b2 = synth.make_pairs_device(lp, g2, dev, seed=synth.SEED + 7 + rank, read_len=L, on_target=0.5)
        torch.cuda.synchronize()
        p2 = [{kk: (v.data_ptr() if v is not None else 0) for kk, v in b2.items()}]
        steps2 = max(2, min(args.steps, 10))
        dt2, tm2, r2, n_assoc2, _, _ = timed(h2, p2, lp, steps2, 1, 1, False)
        k2 = tm2["total_ms"] / max(tm2["n_launches"], 1)
        cfg2 = {"workload": "configs[2] index: 60000 genes (1.78e8 bases, lognormal lengths, every 10th gene shares half of its predecessor), "
                            "%d pairs 2x150 bp per GPU per step, k=17 c=0.6 bf=2^36 bits" % lp,
                "value": round(2 * lp * world * steps2 / dt2, 1), "unit": "reads/s", "n_gpus": world, "steps": steps2,
                "ms_per_step": round(dt2 / steps2 * 1e3, 3), "kernel_ms": round(k2, 4), "probe_mode": h2.probe_mode(), "kernel": h2.last_kernel(),
                "index_build_s": round(t_build2, 3), "n_set_bits": int(info2["n_set_bits"]), "tot_idx": int(info2["tot_idx"]),
                "assoc_per_step": n_assoc2 // steps2, "tie_reads": int(tm2["last_n_tie"])}
        if rank == 0:
            # The ceiling of random memory-side requests is measured here, in this run, with the library's own measurement entry
            # point (same device, a table of the index's size, plain and streaming loads, the better of the two).
            w2 = h2.count_work(lp, p2[0]["seq1"], p2[0]["off1"], p2[0]["seq2"], p2[0]["off2"])
            tab2_bytes = position_table_bytes(int(info2["n_set_bits"]))     # the configs[2] index's position table: 2^29 buckets of 16 bytes = 8 GiB (1.69e8 set bits at load <= 0.3)
            ceil_plain = h2.measure_random_lookups(tab2_bytes, 1 << 31, False)
            ceil_nt = h2.measure_random_lookups(tab2_bytes, 1 << 31, True)
            big["configs2"] = dict(w=w2, kern_ms=k2, n_assoc=n_assoc2 // steps2, ceiling=max(ceil_plain, ceil_nt), hasq=False,
                                   ceiling_measured={"table_bytes": tab2_bytes, "plain": round(ceil_plain, 1), "streaming": round(ceil_nt, 1),
                                                     "how": "shk_measure_random_lookups, this run"})
        if rank == 0 and world == 1:
            # BASELINE configs[2] is "a 60 000-gene FASTA + 100 M reads on one GPU" (README.md:63-66 is the command's shape): the same
            # index at the host boundary (pinned buffers) and through the `shark` command on FASTQ files, 16 M pairs at the north-star
            # read mix, ssv md5 against the device-resident result like every run of that leg
            if not args.no_boundary:
                log("configs[2] index: batch boundary")
                cfg2["batch_boundary"] = boundary_leg(h2, b2, min(lp, 4_000_000), L, kinds=("pinned",), n_batches=8)
            if not args.no_cli:
                log("configs[2] index: the shark command end to end")
                del b2
                b2 = None
                torch.cuda.empty_cache()
                cfg2["cli_end_to_end"] = cli_end_to_end(args, g2, dev, h2, L, ots=[0.50], only_small=True, with_gz=False, with_shared=False,
                                                        extra_args=["-b", "8"], k=17)
        h2.close()
        del b2
        # BASELINE configs[4] shape on this GPU's shard: k=31, -q 20, --single, 2^37-bit filter (the quality-mask path at max k)
        log("configs[4] index")
        t0 = time.time()
        h4 = SharkHip(k=31, c=0.6, bf_bits=1 << 37, min_quality=20, single=True, device=local_rank)
        info4 = h4.build([g.tobytes() for g in g2])
        t_build4 = time.time() - t0
        ceil4 = None
        tab4_bytes = position_table_bytes(int(info4["n_set_bits"]))
        if rank == 0:
            ceil4 = max(h4.measure_random_lookups(tab4_bytes, 1 << 31, False), h4.measure_random_lookups(tab4_bytes, 1 << 31, True))
        old4 = None
        for qual_model in ("uniform", "ends"):
            # "ends": low qualities at the 3' end of the reads (1.2 % of the bases below Q20); "uniform": rounds 1-2's model, 10 % of the
            # bases Q2-29 anywhere, under which a 31-mer survives -q 20 with 0.13 and the hit path is barely exercised (shark_amd/synth.py)
            b4 = synth.make_pairs_device(lp, g2, dev, seed=synth.SEED + 7 + rank, read_len=L, on_target=0.5, with_qual=True, qual_model=qual_model)
            torch.cuda.synchronize()
            p4 = [{kk: (v.data_ptr() if v is not None else 0) for kk, v in b4.items()}]
            dt4, tm4, _, n_assoc4, _, _ = timed(h4, p4, lp, steps2, 1, 1, False)
            k4 = tm4["total_ms"] / max(tm4["n_launches"], 1)
            below_q20 = float((b4["qual1"][:1 << 24] < 53).float().mean().item())
            e4 = {"qualities": qual_model, "frac_bases_below_q20": round(below_q20, 4),
                  "value": round(2 * lp * world * steps2 / dt4, 1), "unit": "reads/s", "ms_per_step": round(dt4 / steps2 * 1e3, 3),
                  "kernel_ms": round(k4, 4), "kernel": h4.last_kernel(), "assoc_per_step": n_assoc4 // steps2, "frac_pairs_assigned": round(n_assoc4 / steps2 / lp, 4)}
            if rank == 0:
                w4 = h4.count_work(lp, p4[0]["seq1"], p4[0]["off1"], p4[0]["seq2"], p4[0]["off2"], p4[0]["qual1"], p4[0]["qual2"])
                big["configs4_" + qual_model] = dict(w=w4, kern_ms=k4, n_assoc=n_assoc4 // steps2, ceiling=ceil4, hasq=True,
                                                     ceiling_measured={"table_bytes": tab4_bytes, "best_of_plain_and_streaming": round(ceil4, 1),
                                                                       "how": "shk_measure_random_lookups, this run, on a table of the size the kernel probes: the index's "
                                                                              "position table (%d set bits at load <= 0.3, two slots per 16-byte bucket) -- the k = 31 "
                                                                              "kernel never reads the 16 GiB filter" % int(info4["n_set_bits"])})
            if qual_model == "uniform":
                old4 = e4
            else:
                cfg4 = {"workload": "configs[4] shape: the same 60000 genes, %d pairs 2x150 bp with qualities (Q30-41, low 3' tails: %.1f %% of the bases "
                                    "below Q20) per GPU per step, k=31 c=0.6 -q 20 --single bf=2^37 bits" % (lp, 100 * below_q20),
                        "n_gpus": world, "steps": steps2, "probe_mode": h4.probe_mode(), "index_build_s": round(t_build4, 3),
                        "n_set_bits": int(info4["n_set_bits"])}
                cfg4.update(e4)
                cfg4["rounds_1_2_workload"] = old4
            del b4, p4
        h4.close()
        del g2
        # the headline context again for the cpu sample / the CLI leg below
        h = SharkHip(k=k, c=c, bf_bits=bf_bits, device=local_rank)
        h.build([g.tobytes() for g in genes])
        res, _ = one_pass(h, ptrs[:1], chunk_pairs)

    # ---- ranks other than 0 are done with their GPU: they leave it alone while rank 0 runs the host-fed leg, and meet rank 0 again
    # in finalize().  They wait on a FILE, not in a collective: a pending RCCL barrier is a kernel spinning on their GPU. ----
    if rank != 0:
        h.close()
        batches.clear()
        torch.cuda.empty_cache()
        t_wait = time.time()
        while not os.path.exists(done_flag) and time.time() - t_wait < 1500:
            time.sleep(0.05)
        sdist.finalize()
        return

    try:
        n = chunk_pairs
        ptr, batch = ptrs[0], batches[0]
        sha = kernel_src_sha()
        # ---- counters: live child passes (rank 0 at N = 1), else the committed file when it matches these kernel sources ----------------
        ctr, ctr_note = {}, None
        want = ["configs1"] + ([] if args.no_configs else ["configs2", "configs4_uniform", "configs4_ends"])
        if world == 1 and not args.no_live_counters and n == lp:
            log("live counter passes (rocprofv3 --pmc children) of %s" % ",".join(want))
            ctr, ctr_note = collect_counters(args, want)
            log(ctr_note)
        missing = [w for w in want if w not in ctr]
        if missing:
            cc, cnote = committed_counters(args)
            for w in missing:
                if w in cc:
                    ctr[w] = dict(cc[w], source="committed")
            ctr_note = (ctr_note + "; " if ctr_note else "") + "for %s: %s" % (",".join(missing), cnote)

        # ---- roofline of the dominant kernel of the headline workload --------------------------------------------------------------
        # SURVEY 8(d)'s exact algorithmic bytes of one launch: bases (+quals) + 8 B per probed k-mer (one 64-bit filter word)
        #   + per hit 8 B rank word + 8 B CSR offsets + 2 B per gene id + 8 B result per read  -> `model_8d`, never `frac`
        log("roofline: exact work counters of one launch")
        w = h.count_work(n, ptr["seq1"], ptr["off1"], ptr["seq2"], ptr["off2"])
        alg_bytes = w["n_bases"] + 8 * w["n_kmers"] + 16 * w["n_hits"] + 2 * w["n_list_ids"] + 8 * (2 * n)
        t_k = kern_ms * 1e-3
        assoc_launch = int(res.n_assoc)
        in_bytes, out_bytes = w["n_bases"], 4 * n + 12 * assoc_launch       # count[] cleared for every read; count + 8 B of inline ids per associated read
        lds_mode = h.probe_mode().startswith("lds-")
        e1 = ctr.get("configs1")
        if lds_mode:
            ip = instruction_part(e1, n, kern_ms) if e1 else None
            if ip:
                roofline = {"bound": "valu-issue", "achieved": ip["G_valu_wave_instructions_per_s"], "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s",
                            "frac": ip["frac_of_valu_issue_peak"],
                            "peak_how": "1024 SIMD-32 x %.1f GHz / 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md)" % CLK_GHZ,
                            "why_this_bound": "the index's exact table sits in LDS (probe mode %s): a pair touches no memory but its own bases; the SQ counters put "
                                              "the kernel's time into VALU issue and exposed LDS latency" % h.probe_mode(),
                            "instructions": ip}
                # ... and against what the SIMDs issue of THIS instruction mix (shk_measure_valu_mix: the kernel's own arithmetic on register
                # operands, nothing to wait for): instructions per wave-iteration from the same counter pass, time measured here
                vm = ctr.get("valu_mix") or {}
                mixc = {}
                for wv in (4, 8):
                    ipi = vm.get(str(wv))
                    if ipi:
                        ms_mix, wi, ghz_mix = h.measure_valu_mix_clock(wv, 20000)
                        mixc["%d_waves_per_simd" % wv] = {"G_valu_wave_instructions_per_s": round(ipi * wi / (ms_mix * 1e-3) / 1e9, 1),
                                                          "valu_per_iteration": round(ipi, 1),
                                                          "simd_cycles_per_valu_instruction": round(ms_mix * 1e-3 * CLK_GHZ * 1e9 * N_SIMD / (ipi * wi), 2),
                                                          "shader_clock_GHz": round(ghz_mix, 3),
                                                          "simd_cycles_per_valu_instruction_at_that_clock": round(ms_mix * 1e-3 * ghz_mix * 1e9 * N_SIMD / (ipi * wi), 2) if ghz_mix else None}
                if mixc:
                    ref = mixc.get("4_waves_per_simd") or list(mixc.values())[0]
                    roofline["mix_ceiling"] = dict(mixc, what="G VALU wave-instructions/s of the exact-table kernel's own instruction mix on register operands "
                                                              "(stage, windows, canonical form, XXH64, table arithmetic, validity / coverage step; no LDS, no memory), "
                                                              "at the kernel's occupancy (4 waves per SIMD: its 144 KiB table admits one 16-wave workgroup per CU) and at 8")
                    roofline["frac_of_mix_ceiling"] = round(ip["G_valu_wave_instructions_per_s"] / ref["G_valu_wave_instructions_per_s"], 4)
                    # the same two readings in CYCLES: the kernel and the mix each at the clock they were measured to hold (the kernel's from
                    # GRBM_GUI_ACTIVE of its counter pass, the mix's from s_memtime / s_memrealtime inside it) -- `frac` prices the kernel at
                    # 2.4 GHz whatever the chip ran at
                    ck, cm = ip.get("simd_cycles_per_valu_instruction_at_that_clock"), ref.get("simd_cycles_per_valu_instruction_at_that_clock")
                    if ck:
                        roofline["clock"] = {"kernel_GHz": ip["shader_clock_GHz"], "mix_GHz": ref.get("shader_clock_GHz"), "nominal_GHz": CLK_GHZ,
                                             "frac_at_measured_clock": ip["frac_of_valu_issue_peak_at_that_clock"],
                                             "frac_of_mix_ceiling_in_cycles": round(cm / ck, 4) if cm else None,
                                             "how": "kernel: GRBM_GUI_ACTIVE / 8 / dispatch duration in one rocprofv3 --pmc pass of this run; mix: sum of "
                                                    "s_memtime deltas / sum of s_memrealtime deltas (100 MHz) around every wave's loop"}
            else:
                roofline = {"bound": "hbm", "achieved": round((in_bytes + out_bytes) / t_k / 1e9, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": round((in_bytes + out_bytes) / t_k / 1e9 / HBM_PEAK_GBPS, 5),
                            "why_this_bound": "no instruction counters for this run (see counters_source): the compulsory HBM bytes (input + results) are what is left "
                                              "to price the kernel against; with counters the line names VALU issue"}
            traffic = None
            if e1:
                fb, wb = fetch_write_bytes(e1)
                if fb is not None:
                    traffic = int(2.0 * fb + (wb or 0.0))
                    roofline["traffic_how"] = "2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, HBM section: FETCH_SIZE tallies 128-B requests at 64 B), per launch"
                    roofline["traffic_counters"] = {"FETCH_SIZE_bytes": int(fb), "WRITE_SIZE_bytes": int(wb or 0), "TCC_MISS_sum": e1.get("TCC_MISS_sum"), "TCC_HIT_sum": e1.get("TCC_HIT_sum")}
                    roofline["traffic_over_compulsory"] = round(traffic / (in_bytes + out_bytes), 3)
            roofline["traffic"] = traffic
            roofline["hbm_compulsory"] = {"bytes": int(in_bytes + out_bytes), "in": int(in_bytes), "out": int(out_bytes), "GBps": round((in_bytes + out_bytes) / t_k / 1e9, 1),
                                          "frac_of_hbm_peak": round((in_bytes + out_bytes) / t_k / 1e9 / HBM_PEAK_GBPS, 4),
                                          "what": "input bases + result bytes of one launch / kernel time / 8 TB/s: what any implementation has to move"}
            roofline["model_8d"] = {"bytes": int(alg_bytes), "bytes_per_read": round(alg_bytes / (2 * n), 1), "GBps": round(alg_bytes / t_k / 1e9, 1),
                                    "x_of_hbm_peak": round(alg_bytes / t_k / 1e9 / HBM_PEAK_GBPS, 4), "kmers": int(w["n_kmers"]), "hits": int(w["n_hits"]),
                                    "why": "SURVEY 8(d)'s figure over EVERY k-mer of every read, as the reference visits them.  Not HBM use and not a roofline: on this index "
                                           "the exact table sits in LDS and the bound cut, the early decision and the sparse first round (DESIGN.md 3) prove most of those "
                                           "probes irrelevant to the read's result and never make them (an off-target pair needs 128 of its 268 k-mers, a pair from the "
                                           "gene 64), so the figure can exceed the HBM peak"}
        else:
            # (another index through --genes / --bf-log2: bound by table lookups -- L2 or memory side)
            ceiling = h.measure_random_lookups(16 << 26, 1 << 30, False)     # (a 1 GiB table: beyond the caches)
            roofline = request_rate_roofline(e1, kern_ms, ceiling, alg_bytes, in_bytes, out_bytes, n, w)
        roofline.update({"kernel": (e1 or {}).get("kernel") or headline_kernel, "kernel_reported_by_library": headline_kernel,
                         "kernel_ms": round(kern_ms, 4), "launches": int(tm["n_launches"]), "pairs_per_launch": n,
                         "kernel_src_sha": sha, "counters_source": ctr_note})

        for name, cfg in (("configs2", cfg2), ("configs4_ends", cfg4), ("configs4_uniform", (cfg4 or {}).get("rounds_1_2_workload"))):
            if cfg is None or name not in big:
                continue
            bb = big[name]
            ww = bb["w"]
            algb = ww["n_bases"] * (2 if bb["hasq"] else 1) + 8 * ww["n_kmers"] + 16 * ww["n_hits"] + 2 * ww["n_list_ids"] + 8 * (2 * lp)
            rl = request_rate_roofline(ctr.get(name), bb["kern_ms"], bb["ceiling"], algb, ww["n_bases"] * (2 if bb["hasq"] else 1), 4 * lp + 12 * bb["n_assoc"], lp, ww,
                                       cal=fetch_calibration(ctr.get("lookup_calibration")))
            rl["ceiling_measured"] = bb["ceiling_measured"]
            rl["counters_source"] = ctr_note
            cfg["roofline"] = rl

        # ---- PCIe-inclusive rate of the host-buffer entry point (never `value`) ---------
        boundary = None
        if not args.no_boundary and world == 1:
            log("batch boundary (host buffers, pipelined)")
            boundary = boundary_leg(h, batch, min(n, 4_000_000), L)
            res, _ = one_pass(h, ptrs[:1], chunk_pairs)      # (`res` again: the boundary leg used the context's result buffers)

        # ---- CPU baseline: the oracle (port of the reference path) on this host -------
        cpu = None
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N=1 only (bounded sample)
            log("cpu baseline (oracle)")
            from oracle import pyoracle
            threads = host_threads()
            model, phys = cpu_info()
            ns = args.cpu_sample_pairs or min(n, threads * 50000)
            # the sample: half from the head of the batch, half from its middle (the generator works in chunks of 2^20 pairs)
            n_head = ns - ns // 2 if n >= 2 * ns else ns
            n_mid, first_mid = ns - n_head, n // 2
            hb = synth.to_host_sample(batch, n_head, L)
            if n_mid:
                hm = synth.to_host_sample(batch, n_mid, L, first=first_mid)
                for kk in ("seq1", "seq2"):
                    hb[kk] = np.concatenate([hb[kk], hm[kk]])
                hb["off1"] = np.arange(0, (ns + 1) * L, L, dtype=np.uint64)
                hb["off2"] = hb["off1"].copy()
            o = pyoracle.Shark(k=k, c=c, bf_bits=bf_bits)
            o.build([g.tobytes() for g in genes])
            t0 = time.perf_counter()
            ogoff, ogids = o.classify(hb["seq1"], hb["off1"], hb["seq2"], hb["off2"], nthreads=threads)
            tc = time.perf_counter() - t0
            # the sample doubles as an end-of-run parity check against the GPU result (`res`: the last launch on this batch)
            parity = True
            for first, cnt, o_first in ((0, n_head, 0), (first_mid, n_mid, n_head)):
                if not cnt:
                    continue
                goff = np.empty(cnt + 1, dtype=np.uint32)
                hip_memcpy_dtoh(goff, res.gene_off + 4 * first, (cnt + 1) * 4)
                n_ids = int(goff[cnt]) - int(goff[0])
                gids = np.empty(max(n_ids, 1), dtype=np.uint16)
                hip_memcpy_dtoh(gids, res.gene_ids + 2 * int(goff[0]), n_ids * 2)
                want_off = ogoff[o_first:o_first + cnt + 1].astype(np.int64) - int(ogoff[o_first])
                parity = parity and bool(np.array_equal(goff.astype(np.int64) - int(goff[0]), want_off)
                                         and np.array_equal(gids[:n_ids], ogids[int(ogoff[o_first]):int(ogoff[o_first + cnt])]))
            # one thread on one reference chunk (SURVEY 8d asks for -t 1 next to all cores)
            n1 = min(n_head, 50000)
            t0 = time.perf_counter()
            o.classify(hb["seq1"][:int(hb["off1"][n1])], hb["off1"][:n1 + 1], hb["seq2"][:int(hb["off2"][n1])], hb["off2"][:n1 + 1], nthreads=1)
            t1 = time.perf_counter() - t0
            cpu = {"value": round(2 * ns / tc, 1), "unit": "reads/s", "cores": threads, "kind": "port",
                   "physical_cores": phys or None, "cpu_model": model,
                   "sample": ("%d pairs of the same batch (%s), %d threads x 50 000-read chunks (main.cpp:215), %.1f s"
                              % (ns, ("the whole launch: head, middle and tail" if ns == n else
                                      "the first %d and %d from its middle" % (n_head, n_mid) if n_mid else "its first %d" % n_head), threads, tc)),
                   "parity_with_gpu": parity,
                   "one_thread": {"value": round(2 * n1 / t1, 1), "unit": "reads/s", "sample": "first %d pairs, %.1f s" % (n1, t1)}}
            o.close()

        # ---- the drop-in command itself, end to end (never `value`): FASTQ files in, ssv + FASTQ files out.  At N > 1: the host-fed
        # multi-GPU leg, `shark --gpus N` (the other ranks have left their GPUs alone by now) ----------------
        cli = None   # (last: it classifies other reads with the same context, which invalidates `res`)
        if not args.no_cli and (world == 1 or sdist.backend_name() == "nccl"):
            for b in batches:
                b.clear()
            batches.clear()
            torch.cuda.empty_cache()
            log("the shark command end to end")
            cli = cli_end_to_end(args, genes, dev, h, L, n_gpus=world)
            if world == 1 and cli.get("runs") is not None:
                # a sample whose 4 M-pair stretches come from the gene and from elsewhere BY TURNS: the library picks a batch's kernel order by
                # what the batch before looked like, and here every such guess is wrong at every turn of the sample
                log("the shark command on a sample of alternating stretches")
                alt = cli_end_to_end(args, genes, dev, h, L, ots=[0.5], only_small=True, with_gz=False, with_shared=False, alternate=4_000_000)
                cli["alternating_stretches"] = {"what": "the same command on %d pairs whose 4 M-pair stretches are drawn from the gene and from elsewhere by turns "
                                                        "(every per-batch guess from the batch before is wrong at every turn)" % args.cli_pairs,
                                                "runs": alt.get("runs"), "value": alt.get("value"), "unit": "reads/s", "all_runs_valid": alt.get("all_runs_valid")}
                if alt.get("runs") is not None and not alt.get("all_runs_valid", True):
                    cli["all_runs_valid"] = False
            log("done")
    finally:
        try:
            open(done_flag, "w").close()
        except OSError:
            pass

    out = {
        "metric": "reads/s (paired 2x150 bp, k=%d)" % k,
        "value": round(value, 1),
        "unit": "reads/s",
        "n_gpus": world,
        "ranks_seen": ranks_seen,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": "configs[1]: %d gene(s) x %d bp, 2x150 bp pairs in launches of %d, k=%d c=%.1f bf=2^%d bits; a read set of %d pairs "
                               "over %d GPU(s) (%s scaling), %d passes over it per step"
                               % (args.genes, args.gene_len, chunk_pairs, k, c, args.bf_log2, pairs_per_pass_all, world, args.scaling, reps),
                   "pairs_per_launch": chunk_pairs, "launches_per_step_per_gpu": len(my_chunks) * reps, "pairs_in_read_set": pairs_per_pass_all,
                   "reps_per_step": reps, "pairs_per_step": pairs_per_pass_all * reps,
                   "reads_per_step": reads_per_step, "on_target": args.on_target,
                   "seed": synth.SEED, "index_build_s": round(t_build, 3), "n_set_bits": int(info["n_set_bits"]),
                   "assoc_per_step": int(n_assoc.item()) // args.steps,
                   "gene_count_checksum": int(np.asarray(counts, dtype=np.uint64).sum()) // args.steps,
                   "gene_counts_allreduce": ("single GPU: no collective" if world == 1 else
                                             "shk_dist_gene_counts_allreduce (RCCL inside libsharkhip; ranks_seen = ncclCommCount); one warm-up collective "
                                             "before the window, the timed one is per_rank[].allreduce_ms" if sdist.backend_name() == "nccl"
                                             else "gloo dry run (SHARK_DIST_BACKEND): torch.distributed all-reduce of the library's local counters"),
                   "timed_window": "barrier, %d steps x %d passes x %d launches per GPU (each shk_classify_device call synchronises once), one all-reduce of "
                                   "uint64[65536], barrier; max over ranks" % (args.steps, reps, len(my_chunks)),
                   "long_reads": int(tm["last_n_long"]), "tie_reads": int(tm["last_n_tie"]), "probe_mode": h.probe_mode()},
        "value_async": value_async,
        "per_rank": per_rank,
        "roofline": roofline,
        "cpu_baseline": cpu,
        "configs": [cfg2, cfg4] if cfg2 else [],
        "trimmed_reads": trimmed,
        "batch_boundary": boundary,
        "cli_end_to_end": cli,
    }
    print(json.dumps(out), flush=True)
    if cli is not None and cli.get("runs") is not None and not cli.get("all_runs_valid", True):
        # a command that printed something else than the library classified must not leave a throughput figure behind unnoticed
        print("bench.py: cli_end_to_end has invalid runs (rc / ssv md5 / FASTQ bytes): see the JSON line", file=sys.stderr)
        sdist.finalize()
        sys.exit(3)
    sdist.finalize()


if __name__ == "__main__":
    main()
