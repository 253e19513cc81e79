#!/usr/bin/env python3
"""bench.py -- k-mers/s processed (count + BFS) at k=31 on MI355X, BASELINE.json's metric.

One "step" = one full pass of the environment-finder hot path over one synthetic read set that is
already resident in HBM: empty the table, count every k-mer occurrence (mc_add_reads_packed_dev),
finalize, then the two BFS passes of --bothdirs False (runBfs(-1), runBfs(+1)) with
--coverage 5 --maxkmers 100000.  Workload = BASELINE.json configs[1]: 10 M x 150 bp reads per GPU
drawn from 10 x 5 Mb random contigs; for --gpus N > 1 every rank holds
its own 10 M reads (weak scaling), keys are exchanged with one RCCL all-to-all bucketed by hash
prefix, thresholded shards are all-gathered and rank 0 runs the BFS.

Prints ONE JSON line on rank 0 (contract in the task brief) with `roofline` (dominant kernel =
the counting kernel, algorithmic bytes per k-mer occurrence from SURVEY.md section 8(d)) and
`cpu_baseline` (oracle/ multi-threaded restatement of the reference's design, timed on the host
cores on a bounded sample; N=1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GENOME_SEED, READ_SEED = 20240531, 42
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=None, help="reads per GPU (default: 10 M; --config 2: 100 M)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--k", type=int, default=None, help="default 31; --config 2: 63")
    ap.add_argument("--contigs", type=int, default=None, help="default 10; --config 2: 100")
    ap.add_argument("--contig-len", type=int, default=5_000_000)
    ap.add_argument("--err", type=int, default=100, help="substitution errors per 10000 bases (E1=100, E0=0)")
    ap.add_argument("--coverage", type=int, default=None, help="default 5; --config 2: 3")
    ap.add_argument("--maxkmers", type=int, default=100000)
    ap.add_argument("--cpu-sample-reads", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--skip-no-hint", action="store_true", help="do not time the counting without a capacity hint (3 extra counting runs)")
    ap.add_argument("--skip-config2", action="store_true", help="do not add the `config2` object (configs[2]'s k = 63 pipeline on the same reads, a few steps)")
    ap.add_argument("--config2-steps", type=int, default=4)
    ap.add_argument("--cpu-baseline-full", action="store_true", help="time the CPU port on the WHOLE workload instead of --cpu-sample-reads (minutes; once a round, into profiles/)")
    ap.add_argument("--capacity-hint", type=int, default=0, help="distinct k-mers per GPU expected (0: estimate from the error rate)")
    ap.add_argument("--config", type=int, default=1, choices=[1, 2],
                    help="BASELINE.json configs[N]: 1 = the headline (10 M reads, k=31, coverage 5, bothdirs False); 2 = k=63 poly-hash "
                         "keys, coverage 3, bothdirs True on 100 x 5 Mb contigs (100 M reads at full size: give --reads to scale down)")
    ap.add_argument("--total-reads", type=int, default=0, help="reads of the whole job, split evenly over the ranks (configs[3]: 1000000000 on 8 GPUs)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import metacherchant_amd as m
    from metacherchant_amd.distributed import ShardedCounter

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    # MC_BENCH_ONE_GPU=1 (tests only): every rank uses cuda:0 and gloo carries the collectives, so that the
    # multi-rank path of this file can be exercised on a 1-GPU box; its numbers mean nothing.
    one_gpu = os.environ.get("MC_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    if args.config == 2:  # BASELINE.json configs[2]
        if args.k is None: args.k = 63
        if args.contigs is None: args.contigs = 100
        if args.coverage is None: args.coverage = 3
        if args.reads is None: args.reads = 100_000_000
    if args.k is None: args.k = 31
    if args.contigs is None: args.contigs = 10
    if args.coverage is None: args.coverage = 5
    if args.reads is None: args.reads = 10_000_000
    if args.total_reads:
        args.reads = args.total_reads // world
    bothdirs = args.config == 2
    k, L, R = args.k, args.read_len, args.reads
    mode = m.KEY_PACKED if k <= 31 else m.KEY_POLY
    n_bases = R * L
    windows = R * (L - k + 1)
    genome_bases = args.contigs * args.contig_len
    # expected distinct keys: the genome's k-mers + a new key for every window with a substitution in it (measured at
    # k = 31, 1 %: 363 M against 372 M; at k = 63: 4.58 G against 4.63 G -- a few windows repeat an error of another read)
    est_distinct = int(min(world * windows, genome_bases + world * windows * (1.0 - (1.0 - args.err / 10000.0) ** k)))
    hint_local = args.capacity_hint or est_distinct // world + (1 << 20)

    ctx = m.Context(k, mode, local_rank, hint_local, m.native.FLAG_SOLID_LIST if world > 1 else 0)
    ctx.set_coverage_hint(args.coverage)  # --coverage is known before the reads are loaded (the CLI does the same)
    # several ranks: rank 0 walks over every rank's table in place (mc_shard_attach); MC_BENCH_WALK=gather keeps round 3's way --
    # the k-mers at or above --coverage gathered into a BFS-only context on rank 0
    walk_in_place = os.environ.get("MC_BENCH_WALK", "shards") != "gather"
    solid = None
    if world > 1 and rank == 0 and not walk_in_place:
        solid = m.Context(k, mode, local_rank, 1 << 20)  # BFS-only (mc_solid_from_pairs_dev): its counting table stays empty

    # ---- synthetic reads straight into HBM (not timed)
    n_words = (n_bases + 31) // 32 + 1
    d_words = torch.empty(n_words, dtype=torch.int64, device=dev)
    d_off = torch.empty(R + 1, dtype=torch.int64, device=dev)
    ctx.synth_reads_dev(GENOME_SEED, args.contigs, args.contig_len, READ_SEED, rank * R, R, L, args.err, d_words, d_off)
    # seed gene: contig 0, bases [100000, 101000)
    seed = m.native.synth_genome(GENOME_SEED, 100000, 1000)
    sv = [0] * (len(seed) - k + 1)
    for i in range(len(sv)):
        v = 0
        for c in seed[i:i + k]:
            v = (v << 2) | int(c)
        sv[i] = v
    seed_hi = np.array([v >> 64 for v in sv], dtype=np.uint64)
    seed_lo = np.array([v & 0xFFFFFFFFFFFFFFFF for v in sv], dtype=np.uint64)

    sc = ShardedCounter(ctx, dev)
    info = {}

    def step():
        t_c = time.perf_counter()
        sc.clear()
        sc.add_reads_dev(d_words, d_off, R, n_bases, windows)
        t_f = time.perf_counter()
        info["distinct"] = sc.finalize()  # (waits for the counting to end)
        info["finalize_s"] = info.get("finalize_s", 0.0) + (time.perf_counter() - t_f)
        info["count_wall_s"] = info.get("count_wall_s", 0.0) + (time.perf_counter() - t_c)
        bctx = ctx
        in_place = world > 1 and walk_in_place and sc.attach_shards(dst=0)
        if world > 1 and walk_in_place and not in_place and "walk_fallback" not in info:
            info["walk_fallback"] = sc.attach_error  # (rank 0 could not map the other ranks' tables: the gather of round 3 instead)
            if rank == 0:
                info["solid_ctx"] = m.Context(k, mode, local_rank, 1 << 20)
        if in_place:
            if "solid" not in info:  # (reported once: every shard's number of k-mers at or above --coverage, from the counter the merge kernel keeps)
                nt = torch.tensor([ctx.export_count(args.coverage)], dtype=torch.int64, device=dev)
                dist.all_reduce(nt)
                info["solid"] = int(nt.item())
        elif world > 1:
            sctx = solid if solid is not None else info.get("solid_ctx")
            if sctx is not None:
                sctx.clear()
            info["solid"] = sc.gather_solid(sctx, args.coverage, dst=0)
            bctx = sctx
        if rank == 0:
            bfs_ms, reached, levels, lookups = 0.0, 0, 0, 0
            # buildEnvironment with bothdirs=False: runBfs(-1), runBfs(+1) -- independent passes, one launch
            jobs = [(seed_hi, seed_lo, 0)] if bothdirs else [(seed_hi, seed_lo, -1), (seed_hi, seed_lo, 1)]
            t_b = time.perf_counter()
            res = bctx.bfs_batch(jobs, args.coverage, args.maxkmers, -1)
            info["bfs_wall_s"] = info.get("bfs_wall_s", 0.0) + (time.perf_counter() - t_b)
            for r in res:
                if r is None:
                    raise SystemExit("BFS found no seed k-mer: synthetic workload broken")
                reached += len(r["lo"])
                levels += r["levels"]
                lookups += r["lookups"]
            bfs_ms = res[0]["device_ms"]
            info.update(bfs_ms=bfs_ms, reached=reached, levels=levels, lookups=lookups)
        if in_place:
            sc.walk_done(dst=0)  # (the other ranks' tables were being read: nobody clears before rank 0 has let go)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync()
    ctx.reset_stats()
    sc.reset_phases()
    info["count_wall_s"] = 0.0
    info["bfs_wall_s"] = 0.0
    info["finalize_s"] = 0.0
    t0 = time.perf_counter()
    step_marks = []
    for _ in range(args.steps):
        step()
        step_marks.append(time.perf_counter())  # (host clock when the step's last call returned: the walk's results are on the host then)
    sync()
    elapsed = time.perf_counter() - t0
    if os.environ.get("MC_BENCH_STEP_TIMES"):  # (debugging: where a slow step sits)
        print("step ms: " + " ".join("%.2f" % (1e3 * (b_ - a_)) for a_, b_ in zip([t0] + step_marks[:-1], step_marks)), file=sys.stderr)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = ctx.stats()
    # what every rank spent where, per step (ms, host clocks): the record of a multi-GPU run must say how many ranks there
    # were and what each of them did, so that a scaling number can be read without the source
    ranks_seen, rank_phases = world, None
    if world > 1:
        ranks_seen = dist.get_world_size()
        mine = torch.tensor([sc.phase_s["extract"], sc.phase_s["exchange_wait"], sc.phase_s["count"], info["finalize_s"],
                             info.get("bfs_wall_s", 0.0), float(sc.bytes_sent), float(st.p1_ms + st.p2_ms + st.p3_ms), float(R)],
                            dtype=torch.float64, device=dev)
        allp = torch.empty(world * mine.numel(), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allp, mine)
        allp = allp.view(world, -1).cpu().tolist()
        rank_phases = [{"rank": r, "reads": int(p[7]), "extract_ms": round(1e3 * p[0] / args.steps, 3), "exchange_wait_ms": round(1e3 * p[1] / args.steps, 3),
                        "count_enqueue_ms": round(1e3 * p[2] / args.steps, 3), "finalize_wait_ms": round(1e3 * p[3] / args.steps, 3),
                        "count_kernels_ms": round(p[6] / args.steps, 3), "walk_ms": round(1e3 * p[4] / args.steps, 3),
                        "exchange_GB_sent": round(p[5] / args.steps / 1e9, 4)} for r, p in enumerate(allp)]

    # The same counting WITHOUT a capacity hint -- how the CLI and the reference's self-growing BigLong2ShortHashMap start:
    # a fresh context per step (64 MB table), which sizes its table from the first level-1 bucket of the batch.
    no_hint = None
    if world == 1 and not args.skip_no_hint and args.config == 1:
        walls = []
        for i in range(4):  # (the first one brings the table's memory into the process: not timed)
            c2 = m.Context(k, mode, local_rank, 0)
            c2.set_coverage_hint(args.coverage)
            torch.cuda.synchronize(dev)
            t_c = time.perf_counter()
            c2.add_reads_packed_dev(d_words, d_off, R, n_bases)
            d2 = c2.finalize()
            dt = time.perf_counter() - t_c
            st2 = c2.stats()
            c2.close()
            if d2 != info["distinct"]:
                raise SystemExit("no-hint context counted %d distinct k-mers, the hinted one %d" % (d2, info["distinct"]))
            if i:
                walls.append(dt)
        no_hint = {"count_wall_ms": 1e3 * sum(walls) / len(walls), "table_grows": int(st2.grows), "table_bytes": int(st2.table_bytes)}

    full = (args.config == 1 and R == 10_000_000) or (args.config == 2 and R == 100_000_000)
    if world > 1 and args.config == 1:
        cfg_label = "configs[3]" if R * world == 1_000_000_000 else ("configs[1] on every GPU (weak scaling; configs[3] = --total-reads 1000000000)" if full else "configs[1] scaled, on every GPU")
    else:
        cfg_label = "configs[%d]%s" % (args.config, "" if full else " scaled")
    out = None
    if rank == 0:
        total_windows = windows * world
        value = total_windows * args.steps / elapsed
        ms_per_step = 1e3 * elapsed / args.steps
        distinct = info["distinct"]
        # algorithmic bytes per k-mer occurrence: packed read bits + key read +
        # count read + count write + key write on first insertion
        A = L / (4.0 * (L - k + 1)) + 12.0 + 8.0 * distinct / float(total_windows)
        # One "launch" = one pass of the counting pipeline over the whole batch (k_p1_extract_scatter ->
        # k_p2_scatter -> k_p3_merge); its duration = the summed HIP-event times of
        # those kernels on the context's stream.  For a direct-kernel run it is k_count_reads alone.
        launches = max(int(st.count_launches), 1)
        avg_ms = st.count_ms / launches if st.count_launches else None
        units_per_launch = st.windows / launches
        roofline = None
        if avg_ms:
            achieved = units_per_launch * A / (avg_ms * 1e-3) / 1e9
            # super-k-mer form of the pipeline (packed keys, k >= 23); with several ranks the level-1 kernel
            # scatters the records (or keys) received from the other ranks instead of extracting them from reads
            sk = mode == m.KEY_PACKED and k >= 23 and os.environ.get("MC_SUPERKMERS") != "0"
            p1 = ("k_sk1w_extract" if world == 1 else "k_sk1_records") if sk else ("k_p1_extract_scatter" if world == 1 else "k_p1_keys_scatter")
            # (the merge kernel of super-k-mer records on one GPU with one region per leaf is k_p3_dedup; the level-2 kernel of a
            # hinted one-GPU run k_sk2_scatter_compact: the names rocprofv3 prints)
            dedup = sk and os.environ.get("MC_P3_DEDUP") != "0" and st.table_slots <= (1 << 21) * 4096
            p3 = "k_p3_dedup" if dedup else "k_p3_merge"
            parts = {p1: st.p1_ms / launches, "k_sk2_scatter" if sk else "k_p2_scatter": st.p2_ms / launches,
                     p3: st.p3_ms / launches}
            if world > 1 and st.binned_runs:  # the binned exchange: the senders did the first level (in `extract`, not in the counting run)
                parts = {"k_sk2_scatter_staged<listed>": st.p2_ms / launches, p3: st.p3_ms / launches}
            if st.long_runs:  # polynomial keys, k > 32, as long records (csrc/count_long.h)
                parts = {"k_skl_extract": st.p1_ms / launches, "k_sk2_scatter_compact<2,2>": st.p2_ms / launches, "k_p3_long": st.p3_ms / launches}
            pipeline = st.p3_ms > 0
            dominant = max(parts, key=parts.get) if pipeline else "k_count_reads"
            # HBM bytes per pipeline run from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH_SIZE doubled
            # as MI355X_MICROARCH.md prescribes), collected by scripts/gpu_pmc.sh for THIS build: the file names the commit
            # its kernels were built from, and the number is only reported while the counting kernels are unchanged since
            # (git diff of csrc/count_pipeline.h and csrc/kmer_device.h against that commit is empty), else null.
            traffic, traffic_source = None, None
            import glob
            pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic_e1.csv")))  # (the newest round's)
            pmc = pmcs[-1] if pmcs else os.path.join(ROOT, "profiles", "r05_pmc_hbm_traffic_e1.csv")
            if pipeline and world == 1 and args.err == 100 and R == 10_000_000 and k == 31 and os.path.exists(pmc):
                import csv
                import subprocess
                rows = [r for r in csv.DictReader(l for l in open(pmc) if not l.startswith("#"))]
                commit = next((l.split()[-1] for l in open(pmc) if l.startswith("# commit")), None)
                fresh = None
                # (no .git on the GPU box -- the file also names a hash of the counting kernels' sources as they were measured)
                measured = next((l.split()[2] for l in open(pmc) if l.startswith("# sources")), None)
                if measured:
                    import hashlib
                    hh = hashlib.sha256()
                    for src in ("count_pipeline.h", "kmer_device.h"):
                        hh.update(open(os.path.join(ROOT, "metacherchant_amd", "csrc", src), "rb").read())
                    fresh = hh.hexdigest()[:16] == measured
                if fresh is None and commit and os.path.isdir(os.path.join(ROOT, ".git")):
                    try:
                        fresh = subprocess.run(["git", "-C", ROOT, "diff", "--quiet", commit, "--", "metacherchant_amd/csrc/count_pipeline.h",
                                                "metacherchant_amd/csrc/kmer_device.h"], timeout=20).returncode == 0
                    except Exception:
                        fresh = None
                if fresh is not False:  # (no git on the box: the file travels with the build it was measured on)
                    gb = 0.0
                    for row in rows:
                        if row.get("dispatch", "1") != "2":
                            continue  # (dispatch 1: the cold step, 2: the warm one -- mc_clear, the same table again: what the bench times)
                        # (k_sk2_scatter is also used by the BFS-table build of a sharded run; on one GPU only by the pipeline)
                        # (the three kernels of the headline's pipeline; the file may also hold the config2 leg's per-window kernels)
                        if row["kernel"].startswith(("mc::k_sk1w_extract", "mc::k_p3_dedup", "mc::k_sk2_scatter", "void mc::k_sk1w_extract", "void mc::k_p3_dedup", "void mc::k_sk2_scatter")):
                            gb += float(row["fetch_GB_corrected_x2"]) + float(row["write_GB"])
                    traffic = round(gb * 1e9)
                    traffic_source = "profiles/%s (commit %s; the warm step's dispatches)" % (os.path.basename(pmc), commit)
            roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
                        "kernel": dominant, "launch": "counting pipeline p1+p2+p3" if pipeline else "k_count_reads",
                        "kernel_ms": {kk: round(v, 3) for kk, v in parts.items()} if pipeline else None,
                        "bytes_per_kmer": round(A, 3), "algorithmic_bytes_per_launch": round(units_per_launch * A),
                        "avg_launch_ms": round(avg_ms, 4), "launches_per_step": launches / args.steps,
                        "count_ms_per_step": round(st.count_total_ms / args.steps, 3)}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args, k, mode, L)
        out = {
            "metric": "k-mers/s processed (count+BFS) at k=%d" % k, "value": value, "unit": "k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64",
            "data": "synthetic",
            "config": {"workload": "%s: %dx%dbp reads per GPU, %dx%d bp random contigs, k=%d%s, coverage=%d, "
                                   "maxkmers=%d, bothdirs=%s, %s" % (
                                       cfg_label,
                                       R, L, args.contigs, args.contig_len, k, " (poly hash keys)" if mode == m.KEY_POLY else "", args.coverage, args.maxkmers, bothdirs,
                                       "E1 1% substitutions" if args.err == 100 else "err=%d/10000" % args.err),
                       "reads_per_gpu": R, "read_len": L, "k": k, "err_per_10k": args.err,
                       "parallelism": "reads sharded x%d, all-to-all of super-k-mer records (keys for k < 23 / hash keys) by owner" % world if world > 1 else "1 GPU"},
            "distinct_kmers": distinct, "bfs": {"ms_per_step": round(info["bfs_ms"], 3), "reached": info["reached"],
                                                 "levels": info["levels"], "lookups": info["lookups"]},
            "table_bytes": int(st.table_bytes), "table_grows": int(st.grows),
            "spilled_records_per_step": int(st.spill_keys) // args.steps, "solid_sweeps_per_step": int(st.solid_sweeps) / args.steps, "solid_list_builds_per_step": int(st.solid_list_builds) / args.steps,
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if st.long_runs:  # hash keys in minimizer bins: what mc_finalize_counts' join of the keys by key cost, and what it found
            out["key_join"] = {"mode": "unchecked (MC_DUP_CHECK=0)" if st.dup_unchecked else "exact: keys joined by key at mc_finalize_counts, the walk's absent look-ups asked again by key (csrc/dup_check.h)",
                               "ms_per_step": round(st.dup_ms / args.steps, 3), "joins_per_step": st.dup_checks / args.steps, "keys_in_several_regions": int(st.dup_keys)}
        out["count_wall_ms"] = round(1e3 * info["count_wall_s"] / args.steps, 3)  # clear + count + finalize as the host sees them (kernels: roofline.count_ms_per_step)
        out["bfs_wall_ms"] = round(1e3 * info.get("bfs_wall_s", 0.0) / args.steps, 3)  # mc_bfs_batch as the host sees it: upload, walk, packed results back, arrays for the caller (kernel: bfs.ms_per_step)
        if no_hint is not None:
            # the headline's context knows the number of distinct k-mers beforehand (capacity hint); these do not
            out["count_ms_no_hint"] = round(no_hint["count_wall_ms"], 3)
            out["value_no_hint"] = total_windows / ((ms_per_step - out["count_wall_ms"] + no_hint["count_wall_ms"]) * 1e-3)
            out["no_hint"] = {"table_grows_per_step": no_hint["table_grows"], "table_bytes": no_hint["table_bytes"],
                              "how": "fresh context (64 MB table) per step; table sized from the first level-1 bucket of the batch; table memory recycled inside the process"}
        if world > 1:
            out["solid_kmers"] = info.get("solid")
            out["walk"] = ("rank 0 reads every rank's counting table in place (mc_shard_attach)" if walk_in_place and "walk_fallback" not in info
                           else "solid k-mers gathered into a BFS-only context on rank 0" + (" (the tables could not be mapped: %s)" % info["walk_fallback"] if "walk_fallback" in info else ""))
            out["alltoall_bytes_sent_rank0_per_step"] = sc.bytes_sent // args.steps
            # the run describes itself: the world size the process group reports (not the flag), which walk ran and why, what
            # travelled and what each rank did
            out["ranks_seen"] = ranks_seen
            out["backend"] = dist.get_backend()
            out["walk_mode"] = "in_place" if walk_in_place and "walk_fallback" not in info else "gather"
            out["walk_fallback"] = info.get("walk_fallback")
            out["exchange_GB_per_step"] = round(sum(p["exchange_GB_sent"] for p in rank_phases), 4)
            out["exchange_chunks"] = sc.n_chunks
            # the form the records travelled in: binned = every owner's records in the order of its run's level-1 buckets (include/mcgpu.h
            # mc_extract_superkmers_binned_dev), the counting run starts at its second level; flat = 16-byte records in any order; keys
            # (the walk's look-ahead: every rank's packed reads brought to the walking rank's store, every record with a pointer into it)
            out["reads_gathered_for_the_walk"] = bool(sc.gather_reads)
            out["exchange_form"] = ("binned (%d fine buckets; %d of %d counting runs from the second level)" % (sc.fine_buckets, st.binned_runs, st.count_launches)
                                    if sc.fine_buckets else ("flat records" if sc.by_minimizer else "keys"))
            out["count_runs_per_step"] = sc.n_count_runs
            out["rank_phases_ms_per_step"] = rank_phases
    if world == 1 and args.config == 1 and not args.skip_config2 and rank == 0:
        # configs[2]'s pipeline (k = 63: polynomial-hash keys, one 12-byte record per window, coverage 3, bothdirs True) on the
        # SAME reads -- 10 M x 150 bp over 10 x 5 Mb is configs[2] scaled to a tenth at its own 30-fold depth --, a few steps:
        # a driver-run k = 63 number with its own roofline fraction beside the headline.
        ctx.close()
        try:
            out["config2"] = config2_leg(args, m, torch, dev, local_rank, d_words, d_off, R, L, n_bases)
        except Exception as e:  # (the headline line must not depend on the extra leg)
            out["config2"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)


def config2_leg(args, m, torch, dev, local_rank, d_words, d_off, R, L, n_bases):
    import numpy as np
    k, cov = 63, 3
    windows = R * (L - k + 1)
    genome_bases = args.contigs * args.contig_len
    est = int(min(windows, genome_bases + windows * (1.0 - (1.0 - args.err / 10000.0) ** k)))
    ctx = m.Context(k, m.KEY_POLY, local_rank, est + (1 << 20))
    ctx.set_coverage_hint(cov)
    seed = m.native.synth_genome(GENOME_SEED, 100000, 1000)
    sv = []
    for i in range(len(seed) - k + 1):
        v = 0
        for c in seed[i:i + k]:
            v = (v << 2) | int(c)
        sv.append(v)
    seed_hi = np.array([v >> 64 for v in sv], dtype=np.uint64)
    seed_lo = np.array([v & 0xFFFFFFFFFFFFFFFF for v in sv], dtype=np.uint64)
    res = {}

    def step():
        ctx.clear()
        ctx.add_reads_packed_dev(d_words, d_off, R, n_bases)
        res["distinct"] = ctx.finalize()
        r = ctx.bfs_batch([(seed_hi, seed_lo, 0)], cov, args.maxkmers, -1)[0]
        if r is None:
            raise RuntimeError("config2: BFS found no seed k-mer")
        res["bfs_ms"], res["reached"] = r["device_ms"], len(r["lo"])

    # Hash keys in minimizer bins (csrc/count_long.h): mc_finalize_counts joins the table's keys by key so that different k-mers
    # with one 64-bit hash share a counter as in the reference (csrc/dup_check.h), and a walk's "absent" look-ups are asked again
    # by key.  `value` is measured with both (the library's default: results do not depend on the table's internal form);
    # `value_unchecked` with MC_DUP_CHECK=0 -- the same kernels without the join, for what the join costs.
    unchecked = None
    if os.environ.get("MC_DUP_CHECK") != "0":
        os.environ["MC_DUP_CHECK"] = "0"
        try:
            step()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(args.config2_steps):
                step()
            torch.cuda.synchronize(dev)
            unchecked = time.perf_counter() - t0
        finally:
            del os.environ["MC_DUP_CHECK"]
    step()
    torch.cuda.synchronize(dev)
    ctx.reset_stats()
    t0 = time.perf_counter()
    for _ in range(args.config2_steps):
        step()
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    st = ctx.stats()
    launches = max(int(st.count_launches), 1)
    A = L / (4.0 * (L - k + 1)) + 12.0 + 8.0 * res["distinct"] / float(windows)
    avg_ms = st.count_ms / launches
    ach = st.windows / launches * A / (avg_ms * 1e-3) / 1e9
    out = {"workload": "configs[2] scaled to a tenth: %dx%dbp reads, %dx%d bp contigs (30-fold), k=63 polynomial-hash keys, coverage=3, bothdirs=True, "
                       "maxkmers=%d, err=%d/10000" % (R, L, args.contigs, args.contig_len, args.maxkmers, args.err),
           "value": windows * args.config2_steps / el, "unit": "k-mers/s", "steps": args.config2_steps, "ms_per_step": round(1e3 * el / args.config2_steps, 3),
           "distinct_kmers": res["distinct"], "bfs": {"ms_per_step": round(res["bfs_ms"], 3), "reached": res["reached"]},
           "table_bytes": int(st.table_bytes),
           "roofline": {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                        "bytes_per_kmer": round(A, 3), "avg_launch_ms": round(avg_ms, 4), "launches_per_step": launches / args.config2_steps,
                        #  `achieved` prices every window at SURVEY 8(d)'s algorithmic bytes -- the reference's own per-window
                        # traffic --, whatever travels: a long-record run moves 32 bytes per ~16 windows between the levels, not 12 a window
                        "bytes_model": "algorithmic, per window (SURVEY.md 8d); what the kernels really move: profiles/r06_pmc_hbm_traffic_config2_scaled.csv",
                        "form": "long records (csrc/count_long.h)" if st.long_runs else "a key per window",
                        "kernel_ms": dict(zip(("k_skl_extract", "k_sk2_scatter_compact<2,2>", "k_p3_long") if st.long_runs else
                                              ("k_p1_extract_scatter", "k_p2_scatter", "k_p3_merge"),
                                              (round(st.p1_ms / launches, 3), round(st.p2_ms / launches, 3), round(st.p3_ms / launches, 3)))),
                        "count_ms_per_step": round(st.count_total_ms / args.config2_steps, 3)}}
    if st.long_runs:
        out["key_join"] = {"mode": "unchecked (MC_DUP_CHECK=0)" if st.dup_unchecked else "exact: keys joined by key at mc_finalize_counts, the walk's absent look-ups asked again by key (csrc/dup_check.h)",
                           "ms_per_step": round(st.dup_ms / args.config2_steps, 3), "joins_per_step": st.dup_checks / args.config2_steps,
                           "keys_in_several_regions": int(st.dup_keys)}
        if unchecked is not None:
            out["value_unchecked"] = windows * args.config2_steps / unchecked
            out["ms_per_step_unchecked"] = round(1e3 * unchecked / args.config2_steps, 3)
    ctx.close()
    return out


def _cpu_full_note():
    """the 1 M-read sample flatters the port (its table fits the last-level caches); the whole-workload figure that
    --cpu-baseline-full measured once a round is quoted beside it (the newest profiles/r*_bench_e1_cpu_full.json)."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_e1_cpu_full.json"))):
        try:
            cb = json.load(open(f)).get("cpu_baseline") or {}
            if cb.get("value"):
                best = (os.path.basename(f), cb["value"], cb.get("cores"))
        except Exception:
            pass
    if not best:
        return ""
    return "; on the WHOLE workload (%s: the table no longer fits the caches) the same port reaches %.1f M k-mers/s on %s threads" % (best[0], best[1] / 1e6, best[2])


def cpu_baseline(args, k, mode, L):
    """The reference's CPU path beside the GPU number, on a bounded sample of the same workload.

    With a JVM and the reference's jar on the box ($MC_REFERENCE_JAR) that is the reference itself:
    `java -jar ... --tool environment-finder -p <cores>` on the sample written as FASTA, timed from its own log
    ("Loading file" .. "Hashtable size" = counting; .. "Finished processing all sequences!" = BFS and output).  This image
    has neither, so normally it is oracle/'s multi-threaded restatement of the reference's design (lock-striped
    sub-maps, 32768-read work items) for the counting phase plus the oracle's BFS: "kind": "port"."""
    import shutil

    import numpy as np

    from oracle import pyoracle as po
    # The sample keeps what decides the reference's cost per k-mer -- read length, depth of coverage, error rate, k,
    # thresholds -- and scales the genome with the reads: n reads over ONE contig of n * L / depth bases (the first n reads
    # of the full set would cover its genome n / reads times as thinly: hardly a k-mer would reach the coverage threshold
    # and the BFS leg would time nothing).
    n = args.reads if args.cpu_baseline_full else min(args.cpu_sample_reads, args.reads)
    depth = args.reads * L / float(args.contigs * args.contig_len)
    glen = max(int(n * L / depth), 400000)
    genome = po.synth_genome(GENOME_SEED, glen)
    reads = po.synth_reads(genome, 1, glen, READ_SEED, 0, n, L, args.err)
    cores = os.cpu_count() or 1
    jar = os.environ.get("MC_REFERENCE_JAR")
    if jar and os.path.exists(jar) and shutil.which("java"):
        try:
            return java_baseline(args, k, L, n, genome, reads, cores, jar)
        except Exception as e:  # (fall through to the port, and say why)
            sys.stderr.write("reference jar run failed (%s): timing the port instead\n" % e)
    words = po.pack(reads)
    off = np.arange(n + 1, dtype=np.uint64) * L
    w, nd, sec, table = po.count_reads_packed_mt(words, off, k, 0 if mode == 0 else mode, cores, want_table=True)
    # the BFS leg on the sample's table (the reference runs one calculator thread per seed: single-threaded per pass)
    seed = genome[100000:101000]
    t0 = time.perf_counter()
    dirs = [0] if args.config == 2 else [-1, 1]
    reached = 0
    for d in dirs:
        r = po.bfs(table, k, 0 if mode == 0 else mode, [seed], d, args.coverage, args.maxkmers, -1)
        reached += 0 if r is None else len(r["lo"])
    bfs_sec = time.perf_counter() - t0
    return {"value": w / (sec + bfs_sec), "unit": "k-mers/s", "cores": cores, "kind": "port",
            "sample": "count + BFS on %d reads of the same kind over one %d-base contig (the workload's %.0f-fold depth; %d k-mer "
                      "occurrences, %d distinct): counting %.2f s on %d threads (%.1f M k-mers/s), BFS %.3f s on one thread "
                      "(%d vertices)%s" % (n, glen, depth, w, nd, sec, cores, w / sec / 1e6, bfs_sec, reached, _cpu_full_note())}


def java_baseline(args, k, L, n, genome, reads, cores, jar):
    import re
    import subprocess
    import tempfile
    from datetime import datetime

    from oracle import pyoracle as po
    with tempfile.TemporaryDirectory() as w:
        with open(os.path.join(w, "reads.fasta"), "w") as f:
            for i in range(n):
                f.write(">r%d\n%s\n" % (i, po.decode(reads[i * L:(i + 1) * L])))
        with open(os.path.join(w, "seed.fasta"), "w") as f:
            f.write(">seed\n%s\n" % po.decode(genome[100000:101000]))
        cmd = ["java", "-jar", jar, "--tool", "environment-finder", "-k", str(k), "--reads", os.path.join(w, "reads.fasta"),
               "--seq", os.path.join(w, "seed.fasta"), "--output", os.path.join(w, "out"), "--work-dir", os.path.join(w, "wd"),
               "-p", str(cores), "--force", "--coverage", str(args.coverage), "--maxkmers", str(args.maxkmers),
               "--bothdirs", "True" if args.config == 2 else "False"]
        t0 = time.perf_counter()
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=3600)
        wall = time.perf_counter() - t0
        stamps = {}
        for line in open(os.path.join(w, "wd", "log"), errors="replace"):
            mt = re.match(r"(\d{4}[.-]\d\d[.-]\d\d[ _]\d\d[:.]\d\d[:.]\d\d)", line)
            if not mt:
                continue
            ts = datetime.strptime(re.sub(r"[._-]", " ", mt.group(1)).replace(":", " "), "%Y %m %d %H %M %S")
            for key in ("Loading file", "Hashtable size", "Finished processing"):
                if key in line and key not in stamps:
                    stamps[key] = ts
        windows = n * (L - k + 1)
        sec = wall
        if "Loading file" in stamps and "Finished processing" in stamps:
            sec = max((stamps["Finished processing"] - stamps["Loading file"]).total_seconds(), 1.0)
        return {"value": windows / sec, "unit": "k-mers/s", "cores": cores, "kind": "reference",
                "sample": "the reference jar (%s) on %d reads of the same kind over one contig at the workload's depth, as FASTA: %.1f s from 'Loading file' to "
                          "'Finished processing all sequences!' (log timestamps, 1 s resolution), %.1f s wall" % (os.path.basename(jar), n, sec, wall)}


if __name__ == "__main__":
    main()
