#!/usr/bin/env python
"""bench.py -- GACT GCUPS on synthetic PacBio-shape reads (BASELINE.json metric).

One step = one pass of the hot path over the rank's shard of candidates:
forward-strand candidates and reverse-complement candidates through the
persistent HIP chain kernel (reads and candidates already resident in HBM),
overlap records back on the host, and for N>1 one RCCL gather of the records
to rank 0.  value = DP cells of all ranks / max-over-ranks time (GCUPS).

N>1 is launched by the driver as
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "darwin-gpu_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

OPS_PER_CELL = 24            # 11 score + 10 pointer + 3 arg-max int32 ops (kept as a secondary figure)
# VALU issue peak of gfx950: four SIMD-32 per CU, a wave64 instruction every 2 cycles = 128 lane-op slots per CU per
# clock (MI355X_MICROARCH.md; the 157.3 TFLOP/s fp32 vector figure is 2 flop x this).  tools/issue_probe.hip
# (profiles/r02/issue_rate_probe.json) measures what a SIMD really issues: v_add_u32 / v_and_b32 / v_bitop3_b32 /
# v_fma_f32 reach 1.5 cycles with eight resident waves, the packed-int16, max, perm and DPP instructions this
# kernel is mostly made of 2.5-2.8 cycles (3.2-3.4 with the three waves its register budget allows).  Round 1
# priced against 4 cycles per instruction (64 lanes per CU per clock); that was the rate of ITS probe's dependent
# pairs on few waves, not the machine's.
NOMINAL_LANES_PER_CU_CLK = 128
ISSUE_CYCLES = 2
FETCH_SIZE_CORRECTION = 2.0


def host_cores():
    """cores this process may really use: cgroup quota if set, else the affinity mask"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except Exception:
        pass
    return n


def pmc_lookup(workload_name, candidates, kernel, cells):
    """Counters of the dominant kernel from the newest committed PMC summary (profiles/r*/pmc_*.json, written by
    tools/pmc_summary.py from scripts/gpu_pmc.sh: separate rocprofv3 --pmc passes of this same command) that was
    taken on the same workload, candidate source, kernel and cell count; None otherwise.  HBM bytes cannot be
    counted from inside this process, and neither can executed instructions."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    if candidates != "dsoft":
        return None
    # (newest round first, within a round the round-end pass before the earlier ones; a fresh clone has one mtime for all)
    paths = sorted(glob.glob(os.path.join(here, "profiles", "r*", "pmc_*%s*.json" % workload_name)),
                   key=lambda q: (os.path.basename(os.path.dirname(q)), "round_end" in os.path.basename(q), os.path.getmtime(q), q), reverse=True)
    for path in paths:
        try:
            d = json.load(open(path))
            b = d["sq"]["_bench"]
            if b["kernel"] != kernel or b["kernel_cells"] != cells:
                continue
            # (per launch: tools/pmc_summary.py averages over the launches of a pass since round 4)
            # the entry of THIS kernel in its plain form (the overlapped sequence's two-set variant, `..., true>`, is another kernel)
            stem = kernel.replace(" ", "").rstrip(">")      # (template arguments this line does not name may follow: <7,13,true>, ...)

            def mine(k):
                k = k.replace(" ", "")
                return k.startswith(stem) and not k.endswith(",true>")
            sq = next(v for k, v in d["sq"].items() if mine(k))
            wr = next(v["WRITE_SIZE"] for k, v in d["write"].items() if mine(k) and "WRITE_SIZE" in v)
            rd = next(v["FETCH_SIZE"] for k, v in d["fetch"].items() if mine(k) and "FETCH_SIZE" in v)
            out = {"source": os.path.relpath(path, here), "insts_valu": sq["SQ_INSTS_VALU"], "active_inst_valu": sq.get("SQ_ACTIVE_INST_VALU"),
                   "gui_active": sq["GRBM_GUI_ACTIVE"], "write_kib": wr, "fetch_kib": rd,
                   "profiled_kernel_ms": b["kernel_ms"]}
            # the seed launch of the same passes (since round 5: its own roofline block)
            try:
                sd = next(v for k, v in d["sq"].items() if k.startswith("seed"))
                out["seed"] = {"insts_valu": sd["SQ_INSTS_VALU"], "gui_active": sd["GRBM_GUI_ACTIVE"],
                               "profiled_kernel_ms": b.get("seed_kernel_ms"), "cells": b.get("seed_kernel_cells")}
            except StopIteration:
                pass
            return out
        except (OSError, KeyError, StopIteration, ValueError):
            continue
    return None


def pmc_default_lookup(workload_name, candidates, cells_per_step):
    """VALU instructions ALL kernels of one step execute, from the newest committed PMC pass of the DEFAULT command
    (profiles/r*/pmc_default_*.json, tools/pmc_default_summary.py over scripts/gpu_pmc_default.sh: rocprofv3 --pmc
    SQ_INSTS_VALU GRBM_GUI_ACTIVE around `bench.py --no-others`, steps in flight as in the timed region); None without one
    for this workload and cell count."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    if candidates != "dsoft":
        return None
    paths = sorted(glob.glob(os.path.join(here, "profiles", "r*", "pmc_default_*%s*.json" % workload_name)),
                   key=lambda q: (os.path.basename(os.path.dirname(q)), "round_end" in os.path.basename(q), os.path.getmtime(q), q), reverse=True)
    for path in paths:
        try:
            d = json.load(open(path))
            if d["cells_per_step"] != cells_per_step:
                continue
            return {"source": os.path.relpath(path, here), "insts_valu_per_step": d["insts_valu_per_step"],
                    "steps_profiled": d["steps_profiled"], "kernels": d["kernels"]}
        except (OSError, KeyError, ValueError):
            continue
    return None


# Algorithmic work of the main launch, per DP cell, in lane-op slots.  One slot = one lane of one
# wave64 VALU instruction; the kernels work on int16 pairs, two cells per slot.  Per cell PAIR the recurrence of
# align.cpp:134-160 needs, in the cheapest formulation known for the scoring in use,
#   affine gaps (any scoring):          11 ops for the scores, 11 more where traceback pointers are made
#   linear gaps (open == extend == mismatch, the reference's params.cfg): 5 and 3 more (gact_lin.hpp: H alone,
#                                                 two of its three maxima in one v_pk_maximum3_f16, op-only pointers,
#                                                 the H_up tag for free since round 3's v_bitop3_b32 form)
# and pointers are needed only inside the window a non-first tile's traceback can reach: early x early of
# tile x tile cells (align.cpp:205), 0.39 at the reference's 320 / 120.  Nothing else is counted: no wavefront
# skew, no loads, no traceback walk, no chain bookkeeping -- those are what `frac` is there to expose.
def floor_slots_per_cell(linear, tile, early, affine_drift=False):
    # (the drifted affine pass, gact_aff.hpp:27,37: 7 instructions per cell pair for the scores -- 8 where mismatch < gap_extend --
    #  and 8 more where pointers are made; priced at 7.5 + 8.  Round 1's tagged affine pass, what the uniform / wide / raw-byte
    #  launches still run: 11 + 11)
    score, pointer = (5, 3) if linear else (7.5, 8) if affine_drift else (11, 11)
    window = (min(early, tile) / tile) ** 2
    return (score + pointer * window) / 2.0, {"score_ops_per_cell_pair": score, "pointer_ops_per_cell_pair": pointer,
                                              "pointer_window_fraction": round(window, 4)}


# the seed launch: first tiles -- arg-max and pointers of the WHOLE tile (align.cpp:173-177, traceback from the arg-max) -- so
# every cell pays the pointer instructions and the arg-max key: linear gaps 5 + 3 + 3 (v_pk_mad key, v_pk_max, and the fold's
# share) per cell pair, affine 7.5 + 8 + 3
def seed_floor_slots_per_cell(linear):
    return ((5 + 3 + 3) if linear else (7.5 + 8 + 3)) / 2.0


PMC_MAX_DRIFT = 0.08         # a committed PMC pass whose kernel time is further than this from the live one is not quoted


def main_kernel_name(st):
    """the main launch's kernel as rocprofv3 names it (template arguments without spaces), from gact_hip_run_stats"""
    name = {"int32": "extend_kernel", "packed16-uniform": "extend_p16_kernel<UniformLayout<20>>",
            "packed16-split": "extend_p16_kernel<SplitLayout<7,13>>",
            "packed16-wide": "extend_p16_kernel<WideLayout>"}[st["layout"]]
    if st.get("linear_gap"):            # linear gap scoring: the drifted pass (gact_lin.hpp)
        name = {"extend_p16_kernel<SplitLayout<7,13>>": "extend_p16_kernel<SplitLayoutLin<7,13>>",
                "extend_p16_kernel<WideLayout>": "extend_p16_kernel<WideLayoutLin>"}[name]
        if st.get("role_waves"):        # DP waves + walker waves (gact_roles.hpp)
            name = name.replace("extend_p16_kernel", "extend_roles_kernel")
        elif st.get("coop_walks"):      # two banks per wave, cooperative batched walks (gact_coop.hpp)
            name = name.replace("extend_p16_kernel", "extend_coop_kernel")
    elif st.get("affine_drift"):        # the drifted affine pass (gact_aff.hpp), split layout
        name = "extend_p16_kernel<SplitLayoutAff<7,13>>"
    elif st["tagged_pointers"]:         # pointer phase on tagged scores: the layouts' TAG variants
        name = {"extend_p16_kernel<UniformLayout<20>>": "extend_p16_kernel<UniformLayout<20,16,true>>",
                "extend_p16_kernel<SplitLayout<7,13>>": "extend_p16_kernel<SplitLayout<7,13,true>>",
                "extend_p16_kernel<WideLayout>": "extend_p16_kernel<WideLayoutTagged>"}[name]
    if st.get("critical_lane"):         # a wide launch beside it holds the longest chains (two kernels: no single PMC line to quote)
        name += " + extend_p16_kernel<WideLayoutLin> (critical lane)"
    return name


def pmc_within_drift(pmc, k_ms):
    """a committed PMC pass is quoted only when the kernel time it was taken at is within PMC_MAX_DRIFT of the live one"""
    prof = pmc.get("profiled_kernel_ms")
    return bool(prof) and abs(prof - k_ms) <= PMC_MAX_DRIFT * k_ms


def short_roofline(st, k_ms, main_cells, info, workload_name, tile, early):
    """roofline of a configuration's main launch, the headline's figures in short"""
    kernel = main_kernel_name(st)
    linear = bool(st.get("linear_gap"))
    floor, _ = floor_slots_per_cell(linear, tile, early, affine_drift=bool(st.get("affine_drift")))
    peak = info["compute_units"] * NOMINAL_LANES_PER_CU_CLK * info["clock_mhz"] * 1e6 / 1e12
    achieved = floor * main_cells / (k_ms * 1e-3) / 1e12
    out = {"bound": "valu", "kernel": kernel, "kernel_ms": round(k_ms, 3), "kernel_cells": int(main_cells),
           "floor_slots_per_cell": round(floor, 3), "achieved": round(achieved, 3), "peak": round(peak, 3),
           "unit": "T lane-op slots/s", "frac": round(achieved / peak, 4),
           "valu_issue_utilisation": None, "executed_slots_per_cell": None, "traffic": None, "source": None}
    pmc = pmc_lookup(workload_name, "dsoft", kernel, main_cells)
    if pmc:
        out["profiled_kernel_ms"] = pmc["profiled_kernel_ms"]
        out["kernel_ms_drift"] = round(k_ms / pmc["profiled_kernel_ms"] - 1.0, 4) if pmc["profiled_kernel_ms"] else None
        if pmc_within_drift(pmc, k_ms):
            out["executed_slots_per_cell"] = round(pmc["insts_valu"] * 64.0 / main_cells, 3)
            out["valu_issue_utilisation"] = round(pmc["insts_valu"] * float(ISSUE_CYCLES) / (info["compute_units"] * 4 * pmc["gui_active"] / 8.0), 4)
            out["traffic"] = int((pmc["write_kib"] + FETCH_SIZE_CORRECTION * pmc["fetch_kib"]) * 1024)
            out["source"] = pmc["source"]
        else:
            out["source"] = "%s refused: taken at %.2f ms, this run %.2f ms (more than %d %% apart)" % (
                pmc["source"], pmc["profiled_kernel_ms"] or 0.0, k_ms, int(PMC_MAX_DRIFT * 100))
    return out


def seed_roofline(st, seed_ms, seed_cells, info, pmc):
    """the seed launch (first tiles: arg-max + pointers of the whole tile) against the same issue peak"""
    linear = bool(st.get("linear_gap"))
    floor = seed_floor_slots_per_cell(linear)
    peak = info["compute_units"] * NOMINAL_LANES_PER_CU_CLK * info["clock_mhz"] * 1e6 / 1e12
    achieved = floor * seed_cells / max(seed_ms * 1e-3, 1e-9) / 1e12
    out = {"bound": "valu", "kernel": {"packed16": "seed_p16_kernel<20>", "int32": "extend_kernel<20>"}[st["seed_layout"]],
           "kernel_ms": round(seed_ms, 3), "kernel_cells": int(seed_cells), "floor_slots_per_cell": round(floor, 3),
           "achieved": round(achieved, 3), "peak": round(peak, 3), "unit": "T lane-op slots/s", "frac": round(achieved / peak, 4),
           "gcups": round(seed_cells / max(seed_ms * 1e-3, 1e-9) / 1e9, 1),
           "valu_issue_utilisation": None, "executed_slots_per_cell": None, "source": None}
    sd = (pmc or {}).get("seed")
    if sd and sd.get("cells") == seed_cells and sd.get("profiled_kernel_ms") and abs(sd["profiled_kernel_ms"] - seed_ms) <= PMC_MAX_DRIFT * seed_ms:
        out["executed_slots_per_cell"] = round(sd["insts_valu"] * 64.0 / seed_cells, 3)
        out["valu_issue_utilisation"] = round(sd["insts_valu"] * float(ISSUE_CYCLES) / (info["compute_units"] * 4 * sd["gui_active"] / 8.0), 4)
        out["source"] = pmc["source"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="ecoli10x")
    ap.add_argument("--candidates", default="dsoft", choices=["dsoft", "synthetic"],
                    help="dsoft: candidates from the D-SOFT filter itself; synthetic: placed from simulator truth")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline sample time")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="use torch.distributed even at N=1")
    ap.add_argument("--no-others", action="store_true",
                    help="skip what a default N=1 run appends: the other single-GPU configurations (pacbio50mb, ont; other_configs) "
                         "and the headline workload off its fastest kernels (variants)")
    ap.add_argument("--slots", type=int, default=4,
                    help="engine slots the timed steps alternate over (steps in flight at once: the fetch of step k is taken after "
                         "step k+S-1 has been launched).  1: every step is launched, waited for and fetched before the next")
    ap.add_argument("--only-variants", action="store_true", help="of what a default N=1 run appends, the variants alone (profiling)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): one genome block of --workload per rank, per-GPU work fixed; strong: the FIXED job "
                         "`--workload config4` (eight pacbio50mb genome blocks, 40,000 reads, 2.67 M candidates) dealt over the N ranks")
    ap.add_argument("--no-config4", action="store_true", help="skip the config4_strong entry every default line carries")
    ap.add_argument("--no-reference-caller", action="store_true",
                    help="skip the reference's own unmodified darwin.cpp on the engine (a child process behind the timed region, N = 1)")
    ap.add_argument("--no-cabi-gather-check", action="store_true", help="N > 1: skip the second gather through the C-ABI's own RCCL path")
    ap.add_argument("--scoring", default="1,-1,-1,-1",
                    help="match,mismatch,gap_open,gap_extend of the headline engine (profiling runs of the affine kernels; the CPU baseline "
                         "and parity gate follow it)")
    ap.add_argument("--strong-blocks-of", default="pacbio50mb",
                    help="the genome blocks the fixed strong-scaling job is made of (tests: a small one; config 4 is pacbio50mb)")
    ap.add_argument("--rehearse-on-one-device", action="store_true",
                    help="the whole N > 1 path -- block exchange, deal, steps in flight, gather, checksums -- with every rank on device 0 "
                         "and the collectives over gloo on host tensors (RCCL refuses two ranks on one device): a rehearsal, not a "
                         "measurement; the line says so")
    args = ap.parse_args()
    if args.workload == "config4":
        args.scaling = "strong"
    if args.scaling == "strong" and args.workload not in ("config4", "ecoli10x"):
        raise SystemExit("bench.py: --scaling strong runs the fixed job `--workload config4`")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.force_dist
    if world != args.gpus and world > 1:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("bench.py: --gpus %d needs one process per GPU: launch with\n  python -m torch.distributed.run "
                         "--nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus %d ..."
                         % (args.gpus, args.gpus, args.gpus))

    torch = dist = None
    if use_dist:
        # torch first: its bundled HIP runtime must be the one the process binds
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        global CDEV
        if args.rehearse_on_one_device:
            CDEV = "cpu"
            local_rank = 0
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world)

    import numpy as np
    from gact_amd import engine, workload
    from gact_amd import dist as gdist

    if args.scaling == "strong":
        # the fixed job of BASELINE config 4 as the headline: same line, "scaling": "strong"
        entry = config4_strong(args, dist, torch, rank, world, local_rank, use_dist, steps=args.steps, warmup=args.warmup, headline=True)
        if rank == 0:
            print(json.dumps(entry))
            sys.stdout.flush()
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- workload: one genome block per rank, read sets replicated everywhere
    t_gen = time.time()
    blk = workload.make_block(args.workload, block=rank, candidates=args.candidates)
    blocks = gdist.exchange_blocks(dist, (blk.rs.reads, blk.cf, blk.cr), world, torch=torch, device=CDEV if use_dist else "cpu")
    reads, cf_all, cr_all = gdist.merge_blocks(blocks)
    my_cf = gdist.deal(cf_all, rank, world)
    my_cr = gdist.deal(cr_all, rank, world)
    from gact_amd import synth
    offs = np.zeros(len(reads) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    cat = np.concatenate(reads)
    rcat = np.concatenate([synth.revcomp(r) for r in reads])
    t_gen = time.time() - t_gen

    global SIDE_SLOTS
    S = SIDE_SLOTS = max(1, args.slots)
    scoring = tuple(int(x) for x in args.scoring.split(","))
    eng = engine.Engine(device_id=local_rank, n_slots=S, scoring=scoring)
    info = eng.device_info()
    eng.upload(engine.SET_REF, cat, offs)
    eng.upload(engine.SET_QUERY, cat, offs)
    eng.upload(engine.SET_QUERY_RC, rcat, offs)
    # forward-strand candidates first, reverse-complement ones after (one launch walks both); the same list in every
    # slot: a step is one pass over it, on whichever slot is its turn
    nf, nr = len(my_cf), len(my_cr)
    for k in range(S):
        eng.candidates_upload(np.concatenate([my_cf, my_cr]), slot=k)

    def barrier():
        for k in range(S):
            eng.sync(k)
        if use_dist:
            _sync_cuda(torch)
            dist.barrier()
            _sync_cuda(torch)

    kernel_ms = []

    # the job's output buffers, owned by the caller, page-locked once, explicitly: fetches are one DMA
    rec_bufs = [np.zeros(nf + nr, dtype=engine.OVERLAP_DTYPE) for _ in range(S)]
    if nf + nr:
        for k in range(S):
            eng.register_output(rec_bufs[k], slot=k)

    gather = None
    if use_dist:
        # the one collective of the path: gather of fixed-size overlap records to rank 0, straight from
        # the engine's device-resident record array (no host round trip in front of RCCL)
        # (32 bytes per record travel: the eight numbers of an output line; dist.LINE_DTYPE)
        gather = gdist.RecordGather(torch, dist, nf + nr, gdist.LINE_BYTES, rank, world, CDEV)
        dev_recs = [gdist.DeviceRecords(eng.device_overlaps_ptr(k), nf + nr, engine.OVERLAP_DTYPE.itemsize) for k in range(S)]

    ran = set()

    def launch(slot):
        ran.add(slot)
        eng.candidates_run_mixed(nf + nr, rc_from=nf, same_file=True, slot=slot)

    def complete(slot, record_ms=False):
        """the output of the step that ran on `slot`, where it is due: records on the host (rank 0's host for N > 1)"""
        gathered = None
        if use_dist:
            eng.sync(slot)                               # the engine's own stream: records complete in HBM
            parts = gather(dev_recs[slot] if CDEV == "cuda" else gdist.lines_from_overlaps(eng.candidates_fetch(nf + nr, slot=slot)))
            if rank == 0:
                gathered = gather.to_host(parts, gdist.LINE_DTYPE)          # the job's output, on the host
            rec = None
        else:
            rec = eng.candidates_fetch(nf + nr, slot=slot, out=rec_bufs[slot])
        if record_ms:
            kernel_ms.append(eng.last_run_stats(slot))
        return rec, gathered

    def run_steps(n_steps, in_flight, record_ms=False):
        """n_steps passes over the candidate list, `in_flight` of them launched before the oldest is completed"""
        last = gdist.run_pipelined(n_steps, in_flight, launch, lambda slot: complete(slot, record_ms))
        return last if last is not None else (None, None)

    # (the caller says it keeps S runs in flight: every launch of the timed region takes the throughput layout, also the
    #  first ones, which find the machine idle -- gact_hip_set_option "runs_in_flight")
    eng.set_option("runs_in_flight", 1 if S > 1 else 0)
    run_steps(args.warmup, S)
    barrier()
    t0 = time.perf_counter()
    rec, gathered = run_steps(args.steps, S)
    barrier()
    dt = time.perf_counter() - t0
    eng.set_option("runs_in_flight", 0)
    # the same steps one at a time (launched, waited for, fetched), as the figure of rounds 1-2: `single_slot`
    n_single, dt_single = args.steps, dt / max(args.steps, 1)
    if S > 1:
        n_single = min(5, max(2, args.steps))
        t1 = time.perf_counter()
        run_steps(n_single, 1)
        barrier()
        dt_single = (time.perf_counter() - t1) / n_single
    single_overlapped = bool(eng.last_run_stats(0)["overlapped_seeding"]) if nf + nr else False
    single_lane = bool(eng.last_run_stats(0).get("critical_lane", False)) if nf + nr else False
    # ... and once more in the engine's plain sequence -- seed launch, then ONE main launch on the whole machine -- for the
    # per-kernel HIP-event times of `roofline`: a kernel's own duration is the time the chip spent on it only when it runs
    # alone (inside the timed region the kernels of S steps share the machine; and one at a time the engine seeds most of a
    # run beside two main launches, gact_hip_run_stats.overlapped_seeding)
    n_roof = min(4, max(2, args.steps))
    eng.set_option("overlap_seed", 0)
    run_steps(1, 1)
    t2 = time.perf_counter()
    run_steps(n_roof, 1, record_ms=True)
    barrier()
    dt_roof = (time.perf_counter() - t2) / n_roof
    eng.set_option("overlap_seed", 1)
    if rec is None:          # distributed run: the engine's full records of this rank, for the cell count and the parity gate
        rec = eng.candidates_fetch(nf + nr, slot=0, out=rec_bufs[0])
        # what rank 0 gathered of EVERY rank is what that rank's engine holds (checksums, one all_gather)
        try:
            rank_sums = gdist.verify_gathered(torch, dist, gdist.lines_from_overlaps(rec), gathered, rank, world, CDEV)
        except RuntimeError as err:
            raise SystemExit("bench.py: %s" % err)
        # the same gather once more through the C-ABI's own RCCL path (gact_hip_comm_*, what host/darwin_hip --rccl-gather
        # uses), outside the timed region, compared on rank 0 with what torch.distributed delivered
        cpp_gather = {"ok": None, "skipped": "--no-cabi-gather-check" if args.no_cabi_gather_check else "gloo rehearsal on one device"} \
            if (args.no_cabi_gather_check or args.rehearse_on_one_device) else \
            cpp_gather_check(eng, dist, torch, rank, world, nf + nr, gathered)
    for k in sorted(ran - {0}):    # every slot that took a step holds the same records
        if eng.candidates_fetch(nf + nr, slot=k).tobytes() != eng.candidates_fetch(nf + nr, slot=0).tobytes():
            raise SystemExit("bench.py: slot %d's records differ from slot 0's" % k)
    rf, rr = rec[:nf], rec[nf:]

    my_cells = int(rf["cells"].sum() + rr["cells"].sum())
    my_tiles = int(rf["n_tiles"].sum() + rr["n_tiles"].sum())
    tot_cells, max_dt = my_cells, dt
    if use_dist:
        v = torch.tensor([float(my_cells), float(my_tiles)], dtype=torch.float64, device=CDEV)
        dist.all_reduce(v)
        tot_cells, tot_tiles = int(v[0].item()), int(v[1].item())
        m = torch.tensor([dt], dtype=torch.float64, device=CDEV)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        max_dt = float(m.item())
    else:
        tot_tiles = my_tiles

    gen_all = None
    if use_dist:
        g = [torch.zeros(1, dtype=torch.float64, device=CDEV) for _ in range(world)]
        dist.all_gather(g, torch.tensor([t_gen], dtype=torch.float64, device=CDEV))
        gen_all = [round(float(x.item()), 1) for x in g]
    if rank == 0:
        gcups = tot_cells * args.steps / max_dt / 1e9
        # dominant kernel: the main launch (packed-int16 extend_p16_kernel, or the int32 extend_kernel when
        # the scoring does not fit int16); HIP events on its own stream, this rank
        packed = kernel_ms[-1]["packed16"]
        k_ms = float(np.mean([k["main_ms"] for k in kernel_ms]))
        seed_ms = float(np.mean([k["seed_ms"] for k in kernel_ms]))
        seed_cells = int(kernel_ms[-1]["seed_cells"])
        main_cells = my_cells - seed_cells
        achieved_tops = OPS_PER_CELL * main_cells / (k_ms * 1e-3) / 1e12
        main_kernel = main_kernel_name(kernel_ms[-1])
        measured_rate = eng.measure_valu_rate()
        # issue peak: one wave64 VALU instruction per SIMD per 2 cycles = 128 lane-op slots per CU per clock
        peak_slots = info["compute_units"] * NOMINAL_LANES_PER_CU_CLK * info["clock_mhz"] * 1e6 / 1e12
        linear = bool(kernel_ms[-1].get("linear_gap"))
        floor, model = floor_slots_per_cell(linear, eng.tile_size, eng.tile_size - eng.tile_overlap,
                                            affine_drift=bool(kernel_ms[-1].get("affine_drift")))
        achieved = floor * main_cells / (k_ms * 1e-3) / 1e12
        pmc = pmc_lookup(args.workload, args.candidates, main_kernel, main_cells)
        executed = traffic = traffic_source = None
        pmc_refused = None
        if pmc and not pmc_within_drift(pmc, k_ms):
            # the committed counters were taken at another kernel time (another build, another box): not quoted as this run's
            pmc_refused = "%s refused: taken at %.2f ms, this run %.2f ms (more than %d %% apart)" % (
                pmc["source"], pmc["profiled_kernel_ms"] or 0.0, k_ms, int(PMC_MAX_DRIFT * 100))
            pmc_seed, pmc = pmc, None
        else:
            pmc_seed = pmc
        if pmc:
            slots = pmc["insts_valu"] * 64.0 / main_cells
            # GRBM_GUI_ACTIVE counts over the 8 XCDs; 1024 SIMDs, 2 cycles per issue at the peak
            util = pmc["insts_valu"] * float(ISSUE_CYCLES) / (info["compute_units"] * 4 * pmc["gui_active"] / 8.0)
            # SQ_ACTIVE_INST_VALU counts, per wave, the quad-cycles (4 cycles) in which a VALU instruction of that wave is executing;
            # summed over a SIMD's waves against the launch's cycles it says how much of the time the SIMD's vector pipe was taken
            # -- the reading under which this launch is issue-bound, beside `valu_issue_utilisation`'s 2-cycle one
            pipe = (pmc["active_inst_valu"] * 4.0 / (info["compute_units"] * 4 * pmc["gui_active"] / 8.0)) if pmc.get("active_inst_valu") else None
            executed = {"slots_per_cell": round(slots, 3), "valu_issue_utilisation": round(util, 4),
                        "valu_pipe_occupancy": round(pipe, 4) if pipe else None,
                        "floor_over_executed": round(floor / slots, 4), "profiled_kernel_ms": pmc["profiled_kernel_ms"],
                        # counters cannot be read from inside this process: they are the committed rocprofv3 --pmc pass of this
                        # command; how far this run's kernel time is from the one they were taken at
                        "kernel_ms_drift": round(k_ms / pmc["profiled_kernel_ms"] - 1.0, 4) if pmc["profiled_kernel_ms"] else None,
                        "source": pmc["source"] + " (SQ_INSTS_VALU, GRBM_GUI_ACTIVE of this kernel on this workload; "
                                  "utilisation = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8))"}
            # MI355X_MICROARCH.md: on gfx950 FETCH_SIZE under-reports wide (16-byte) streaming reads by 2x -- the walker's
            # region refills are exactly that -- so the read side is doubled; WRITE_SIZE is taken as reported
            traffic = int((pmc["write_kib"] + FETCH_SIZE_CORRECTION * pmc["fetch_kib"]) * 1024)
            traffic_source = pmc["source"] + " (WRITE_SIZE / FETCH_SIZE in separate passes; %.1f GB written as reported + " \
                "%.1f GB read = %gx the reported FETCH_SIZE, the guide's gfx950 correction for 16-byte reads)" % (
                    pmc["write_kib"] * 1024 / 1e9, FETCH_SIZE_CORRECTION * pmc["fetch_kib"] * 1024 / 1e9, FETCH_SIZE_CORRECTION)
        roofline = {
            # integer-VALU roofline.  The two figures that do not depend on a model of the algorithm
            # come first: what share of the SIMDs' issue slots the kernel used, and how many lane-op slots it executed
            # per DP cell (both from the committed PMC pass of this kernel on this workload; null without one).
            # achieved = algorithmic lane-op slots (model below) x cells of the main launch / its HIP-event time;
            # peak = the SIMDs' issue rate; frac <= 1 by construction = (floor / executed slots per cell) x utilisation.
            # The floor is the cheapest formulation KNOWN for the scoring in use and has moved between rounds
            # (4.87 -> 3.78 -> 3.28 -> 3.09 slots per cell): compare rounds on valu_issue_utilisation, executed_slots_per_cell
            # and survey_24op_int32, not on frac.
            "bound": "valu",
            # which execution mode the figures of this block describe: the main launch alone, one step at a time -- the
            # `single_slot` leg, where a kernel's HIP-event duration is the time the chip spent on it.  `pipelined` below
            # describes the timed region `value` is measured in.
            "mode": "one step at a time, seed launch then one main launch (%d steps behind the timed region; "
                    "%.3f ms per step, %.1f GCUPS)" % (n_roof, dt_roof * 1e3, my_cells / dt_roof / 1e9),
            "valu_issue_utilisation": executed["valu_issue_utilisation"] if executed else None,
            "valu_pipe_occupancy": executed["valu_pipe_occupancy"] if executed else None,
            "executed_slots_per_cell": executed["slots_per_cell"] if executed else None,
            "achieved": round(achieved, 3), "peak": round(peak_slots, 3),
            "unit": "T lane-op slots/s (one lane of one wave64 VALU instruction; int16 pairs: two DP cells per slot)",
            "frac": round(achieved / peak_slots, 4),
            "model": dict(model, slots_per_cell=round(floor, 3), scoring="linear gaps" if linear else "affine gaps"),
            "executed": executed, "pmc_refused": pmc_refused,
            "traffic": traffic, "traffic_source": traffic_source,
            "kernel": main_kernel,
            "measured_in": "%d steps run one at a time in the plain sequence right after the timed region (see `mode`); "
                           "`pipelined` describes the timed region itself" % n_roof,
            "kernel_ms": round(float(k_ms), 3),
            "kernel_cells": main_cells, "seed_kernel": {"packed16": "seed_p16_kernel<20>", "int32": "extend_kernel<20>"}[kernel_ms[-1]["seed_layout"]],
            "seed_kernel_ms": round(seed_ms, 3), "seed_kernel_cells": seed_cells,
            "measured_packed16_issue_tops": round(measured_rate / 1e12, 3),
            # SURVEY.md 8d's first-order figure, kept for continuity: 24 int32 ops per cell against the issue peak.
            # The kernel does not execute that work (two cells per lane-op, pointers only inside the window, arg-max
            # only in first tiles), so this is not a roofline fraction.
            "survey_24op_int32": {"achieved_tops": round(OPS_PER_CELL * main_cells / (k_ms * 1e-3) / 1e12, 3),
                                  "ratio_to_issue_peak": round(OPS_PER_CELL * main_cells / (k_ms * 1e-3) / 1e12 / peak_slots, 4),
                                  "peak_gcups_at_24_ops": round(peak_slots * 1e3 / OPS_PER_CELL, 1)},
            "compute_units": info["compute_units"], "clock_mhz": info["clock_mhz"],
            "waves_per_cu": info["waves_per_cu"],
            # the seed launch against the same peak (every kernel that is 5 % of a step or more has a frac in this line)
            "seed": seed_roofline(kernel_ms[-1], seed_ms, seed_cells, info, pmc_seed),
        }
        # ---- the same roofline for what `value` measured: every kernel of a step (seed, main, ordering, gather) against
        #      the wall time of a step with S steps in flight.  Algorithmic slots: the floor above over ALL cells of a step
        #      (the seed launch's first tiles included, priced like the rest); executed slots and utilisation from the
        #      committed PMC pass of this very command, counters summed over all kernels and divided by the steps it ran.
        step_s = max_dt / args.steps
        pipe_achieved = floor * my_cells / step_s / 1e12
        pmc_d = pmc_default_lookup(args.workload, args.candidates, my_cells) if not use_dist else None
        roofline["pipelined"] = {
            "mode": "%d step(s) in flight: the timed region `value` is measured in" % S,
            "ms_per_step": round(step_s * 1e3, 3), "cells_per_step": my_cells,
            "achieved": round(pipe_achieved, 3), "peak": round(peak_slots, 3), "frac": round(pipe_achieved / peak_slots, 4),
            "executed_slots_per_cell": round(pmc_d["insts_valu_per_step"] * 64.0 / my_cells, 3) if pmc_d else None,
            # SIMD issue slots of the whole chip over the wall time of a step, at the nominal clock
            "valu_issue_utilisation": round(pmc_d["insts_valu_per_step"] * float(ISSUE_CYCLES) /
                                            (info["compute_units"] * 4 * info["clock_mhz"] * 1e6 * step_s), 4) if pmc_d else None,
            "source": (pmc_d["source"] + " (SQ_INSTS_VALU of every kernel of %d profiled steps / steps; utilisation = that x 2 cycles / "
                       "(1024 SIMDs x %d MHz x ms_per_step))" % (pmc_d["steps_profiled"], info["clock_mhz"])) if pmc_d else None,
        }
        out = {
            "metric": "GACT GCUPS (DP cells/s) on ~10 kb PacBio-shape reads",
            "value": round(gcups, 2), "unit": "GCUPS", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(max_dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("int16x2 (packed)" if kernel_ms[-1]["seed_layout"] == "packed16"
                      else "int16x2 (packed) main kernel, int32 seed kernel") if packed else "int32",
            "data": "synthetic",
            "config": {"workload": args.workload + "_self_overlap", "candidate_source": args.candidates, "tile_size": 320, "tile_overlap": 120,
                       "scoring": "%+d/%+d/%+d/%+d" % scoring, "reads": len(reads), "bases": int(offs[-1]),
                       "candidates": int(len(cf_all) + len(cr_all)), "tiles": tot_tiles,
                       "cells_per_step": tot_cells, "parallelism": "candidates dealt round-robin over %d GPU(s)" % world,
                       # steps in flight: step k's records are fetched after step k+S-1 has been launched, each step on an engine
                       # slot of its own (stream, queues, workspace) -- what the reference's 8 feeder threads with a GPU_storage
                       # each do (darwin.cpp:619-629); every step is complete, records on the host, inside the timed region
                       "slots_in_flight": S,
                       # every pass over the candidate list this process made (warm-up, timed region, the one-at-a-time leg, the
                       # roofline leg and its warm-up): what a profile of this command has to be divided by
                       "passes_in_this_process": args.warmup + args.steps + (n_single if S > 1 else 0) + 1 + n_roof,
                       "arch": info["arch"], "gen_seconds": round(t_gen, 1)},
            "roofline": roofline,
            # the same pass one step at a time (launch, wait, fetch, then the next): the figure of rounds 1 and 2
            "single_slot": {"value": round(my_cells * (world if use_dist else 1) / dt_single / 1e9, 2) if not use_dist else None,
                            "ms_per_step": round(dt_single * 1e3, 3), "steps": n_single,
                            # (one at a time the engine seeds a run of this size in length order, most of it beside the main
                            #  launch: gact_hip_run_stats.overlapped_seeding)
                            "overlapped_seeding": single_overlapped,
                            # (... and a wide launch beside the split one takes the longest chains: gact_hip_run_stats.critical_lane)
                            "critical_lane": single_lane},
        }
        if gathered is not None:
            # what the one RCCL gather delivered: records per rank and each rank's own checksum of what it sent (compared on
            # rank 0 with what arrived, dist.verify_gathered), the seconds every rank took to build its block
            out["config"]["gathered_records"] = int(sum(len(g) for g in gathered))
            if args.rehearse_on_one_device:
                out["config"]["rehearsal"] = "every rank on device 0, collectives over gloo on host tensors: the N > 1 code path, not a measurement"
            out["config"]["gather"] = {"ranks": world, "records_per_rank": [int(len(g)) for g in gathered],
                                       "crc32_per_rank": rank_sums if isinstance(rank_sums, list) else None,
                                       "gen_seconds_per_rank": gen_all, "c_abi_rccl_gather": cpp_gather}

        if not args.no_cpu:
            out["cpu_baseline"], out["parity"] = cpu_baseline(args, cat, offs, rcat, my_cf, rf, my_cr, rr, scoring)
            if scoring == (1, -1, -1, -1):
                out["cpu_baseline"]["reference"] = reference_baseline(reads, my_cf, rf)
        if world == 1 and not use_dist and not args.no_others and args.workload == "ecoli10x" and scoring == (1, -1, -1, -1):
            # the other single-GPU configurations of BASELINE.json, a few steps each, with their own parity gate
            eng.close()
            eng = None
            if not args.only_variants:
                # BASELINE config 2 as the reference runs it: 8 feeder threads, each with its own slot and an eighth of the list
                out["feeder_threads"] = feeder_config(args.workload, cat, offs, rcat, my_cf, my_cr, rec)
                out["other_configs"] = [side_config(w, args) for w in ("pacbio50mb", "ont")]
            out["variants"] = [variant_config(v, args.workload, reads, my_cf, my_cr) for v in VARIANTS]
    # ---- the fixed eight-block job of BASELINE config 4 dealt over these N ranks (strong scaling): every default line carries it
    if not args.no_config4 and not args.only_variants and not _CPP_GATHER_STUCK and args.workload == "ecoli10x":
        if eng is not None:
            eng.close()
            eng = None
        entry = config4_strong(args, dist, torch, rank, world, local_rank, use_dist, steps=3, warmup=1, headline=False)
        if rank == 0:
            out["config4_strong"] = entry
    if rank == 0:
        if world == 1 and not use_dist and not args.no_reference_caller and args.workload == "ecoli10x" and not args.only_variants:
            if eng is not None:
                eng.close()
                eng = None
            out["reference_caller"] = reference_caller()
        print(json.dumps(out))
        sys.stdout.flush()
    if use_dist and _CPP_GATHER_STUCK:
        # a collective of the C-ABI check never came back on SOME rank (every rank knows: the flag was all-reduced): the
        # check's thread cannot be joined and the communicator is not to be trusted -- the line is out, every rank leaves
        # the same way, and not with a success code
        os._exit(3)
    if eng is not None:
        eng.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


_CPP_GATHER_STUCK = False
CDEV = "cuda"            # where the collectives' tensors live: "cuda" under nccl (= RCCL), "cpu" under the gloo rehearsal


def _sync_cuda(torch):
    if CDEV == "cuda":
        torch.cuda.synchronize()


def _config4_build_one(job):
    from gact_amd import workload
    b, name = job
    return workload.config4_blocks([b], name=name)[0]


def config4_strong(args, dist, torch, rank, world, local_rank, use_dist, steps, warmup, headline):
    """BASELINE config 4 as ONE node runs it, and as a strong-scaling job: the FIXED set of eight pacbio50mb genome blocks
    (40,000 reads of ~10 kb, 418 Mb, three resident sets on every GPU; 2.67 M D-SOFT candidates, 8.2e12 cells per step),
    the merged candidate list dealt round-robin over the N ranks, one gather of 32-byte lines to rank 0.
    Steps, barriers and the max-over-ranks clock as in the headline.  The reference has nothing to mirror
    (cuda_host.cu:195 hard-wires device 0); the join is README:25's `cat | sort`.
    Parity inside the run: the ranks that hold rank 0's / rank 5's share OF EIGHT compare every one of those records
    with the oracle's golden records (tests/golden/config_config4_rank{0,5}.npz)."""
    import zlib
    import numpy as np
    from gact_amd import engine, workload, synth
    from gact_amd import dist as gdist
    NB = workload.CONFIG4_BLOCKS
    t0 = time.time()
    mine = list(range(rank, NB, world))
    if world == 1 and not use_dist:
        # one process builds all eight: four at a time (each filter holds a 1 GiB index and a quarter of the cores)
        # (fresh interpreters: this process has a HIP runtime in it by now)
        import multiprocessing
        from concurrent.futures import ProcessPoolExecutor
        saved = os.environ.get("LOCAL_WORLD_SIZE")
        os.environ["LOCAL_WORLD_SIZE"] = "4"
        try:
            with ProcessPoolExecutor(4, mp_context=multiprocessing.get_context("spawn")) as pool:
                built = dict(pool.map(_config4_build_one, [(b, args.strong_blocks_of) for b in mine]))
        finally:
            if saved is None:
                os.environ.pop("LOCAL_WORLD_SIZE", None)
            else:
                os.environ["LOCAL_WORLD_SIZE"] = saved
    else:
        built = dict(workload.config4_blocks(mine, candidates=args.candidates, name=args.strong_blocks_of))
    blocks = gdist.exchange_block_rounds(dist, built, NB, rank, world, torch=torch, device=CDEV if use_dist else "cpu")
    reads, cf_all, cr_all = gdist.merge_blocks(blocks)
    offs = np.zeros(len(reads) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    cat = np.concatenate(reads)
    rcat = np.concatenate([synth.revcomp(r) for r in reads])
    my_cf, my_cr = gdist.deal(cf_all, rank, world), gdist.deal(cr_all, rank, world)
    nf, nr = len(my_cf), len(my_cr)
    t_build = time.time() - t0

    S = max(1, args.slots)
    eng = engine.Engine(device_id=local_rank, n_slots=S)
    info = eng.device_info()
    eng.upload(engine.SET_REF, cat, offs); eng.upload(engine.SET_QUERY, cat, offs); eng.upload(engine.SET_QUERY_RC, rcat, offs)
    cands = np.concatenate([my_cf, my_cr])
    bufs = [np.zeros(nf + nr, dtype=engine.OVERLAP_DTYPE) for _ in range(S)]
    for k in range(S):
        eng.candidates_upload(cands, slot=k)
        eng.register_output(bufs[k], slot=k)
    gather = dev_recs = None
    if use_dist:
        gather = gdist.RecordGather(torch, dist, nf + nr, gdist.LINE_BYTES, rank, world, CDEV)
        dev_recs = [gdist.DeviceRecords(eng.device_overlaps_ptr(k), nf + nr, engine.OVERLAP_DTYPE.itemsize) for k in range(S)]
    gather_s = [0.0]

    def barrier():
        for k in range(S):
            eng.sync(k)
        if use_dist:
            _sync_cuda(torch); dist.barrier(); _sync_cuda(torch)

    def launch(slot):
        eng.candidates_run_mixed(nf + nr, rc_from=nf, same_file=True, slot=slot)

    def complete(slot):
        if use_dist:
            eng.sync(slot)
            tg = time.perf_counter()
            parts = gather(dev_recs[slot] if CDEV == "cuda" else gdist.lines_from_overlaps(eng.candidates_fetch(nf + nr, slot=slot)))
            out = gather.to_host(parts, gdist.LINE_DTYPE) if rank == 0 else None
            gather_s[0] += time.perf_counter() - tg
            return None, out
        return eng.candidates_fetch(nf + nr, slot=slot, out=bufs[slot]), None

    eng.set_option("runs_in_flight", 1 if S > 1 else 0)
    gdist.run_pipelined(warmup, S, launch, complete)
    barrier()
    gather_s[0] = 0.0
    t1 = time.perf_counter()
    rec, gathered = gdist.run_pipelined(steps, S, launch, complete) or (None, None)
    eng.sync(0)
    my_dt = time.perf_counter() - t1                                # this rank's own steps (its last gather included)
    barrier()
    dt = time.perf_counter() - t1
    eng.set_option("runs_in_flight", 0)
    # one run at a time on this rank (what a caller that submits its list once sees), outside the timed region
    t2 = time.perf_counter()
    launch(0)
    single_rec = eng.candidates_fetch(nf + nr, slot=0, out=bufs[0]).copy()
    dt_single = time.perf_counter() - t2
    st = eng.last_run_stats(0)
    rec = single_rec
    my_cells, my_tiles = int(rec["cells"].sum()), int(rec["n_tiles"].sum())
    # ---- parity: rank g of EIGHT's share is the golden file's; here it lies on rank g % world, every (8 / world)-th record
    checked = {}
    for g in (0, 5):
        path = os.path.join(ROOT, "tests", "golden", "config_config4_rank%d.npz" % g)
        if NB % world or g % world != rank or not os.path.exists(path) or args.strong_blocks_of != "pacbio50mb":
            continue
        gold = np.load(path)
        stride, first = NB // world, g // world
        got = np.concatenate([rec[:nf][first::stride], rec[nf:][first::stride]])
        sub = np.concatenate([my_cf[first::stride], my_cr[first::stride]])
        if int(gold["candidates_crc"]) != zlib.crc32(sub.tobytes()):
            raise SystemExit("bench.py: tests/golden/config_config4_rank%d.npz was made for another candidate list" % g)
        bad = np.flatnonzero(workload.record_crcs(got) != gold["crc"])
        if len(bad):
            raise SystemExit("bench.py: PARITY FAILURE, config 4: %d records of rank %d of 8 differ from the oracle's golden records, first %s"
                             % (len(bad), g, got[bad[0]]))
        checked["rank%d_of_8" % g] = int(len(got))
    tot_cells, tot_tiles, max_dt = my_cells, my_tiles, dt
    per_rank_ms, per_rank_single, gather_ms, all_checked, builds = [my_dt / steps * 1e3], [dt_single * 1e3], None, dict(checked), [round(t_build, 1)]
    if use_dist:
        v = torch.tensor([float(my_cells), float(my_tiles)], dtype=torch.float64, device=CDEV)
        dist.all_reduce(v)
        tot_cells, tot_tiles = int(v[0].item()), int(v[1].item())
        m = torch.tensor([dt], dtype=torch.float64, device=CDEV)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        max_dt = float(m.item())
        row = torch.tensor([my_dt / steps * 1e3, dt_single * 1e3, float(sum(checked.values())), t_build, gather_s[0] / steps * 1e3],
                           dtype=torch.float64, device=CDEV)
        rows = [torch.zeros_like(row) for _ in range(world)]
        dist.all_gather(rows, row)
        rows = [[float(x) for x in r.tolist()] for r in rows]
        per_rank_ms = [r[0] for r in rows]; per_rank_single = [r[1] for r in rows]; builds = [round(r[3], 1) for r in rows]
        all_checked = {"records_checked_against_golden_all_ranks": int(sum(r[2] for r in rows))}
        gather_ms = round(rows[0][4], 3)
        # what rank 0 gathered of every rank is what that rank's engine holds
        try:
            gdist.verify_gathered(torch, dist, gdist.lines_from_overlaps(rec), gathered, rank, world, CDEV)
        except RuntimeError as err:
            raise SystemExit("bench.py: config 4: %s" % err)
    eng.close()
    if rank != 0:
        return None
    gcups = tot_cells * steps / max_dt / 1e9
    entry = {
        "workload": "config4_fixed_job (8 %s genome blocks, self-overlap)" % args.strong_blocks_of, "scaling": "strong",
        "value": round(gcups, 2), "unit": "GCUPS", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(max_dt / steps * 1e3, 3), "slots_in_flight": S,
        "reads": len(reads), "bases": int(offs[-1]), "candidates": int(len(cf_all) + len(cr_all)), "tiles": tot_tiles, "cells_per_step": tot_cells,
        "parallelism": "the merged candidate list dealt round-robin over %d GPU(s), read sets replicated" % world,
        # every rank's own time per step (its steps launched, waited for, its part of the gather done) and the one gather as
        # rank 0 sees it (the wait for the slowest rank included)
        "per_rank_ms_per_step": {"min": round(min(per_rank_ms), 3), "max": round(max(per_rank_ms), 3), "all": [round(x, 3) for x in per_rank_ms]},
        "gather_ms_per_step_rank0": gather_ms,
        "single_run_ms_per_rank": {"min": round(min(per_rank_single), 2), "max": round(max(per_rank_single), 2)},
        "kernel_layout": st["layout"] + ("-lin" if st["linear_gap"] else "") + ("-roles" if st.get("role_waves") else "-coop" if st.get("coop_walks") else ""),
        "kernel_ms": round(st["main_ms"], 3), "seed_kernel_ms": round(st["seed_ms"], 3),
        "parity": dict(all_checked, bit_exact=True, golden="tests/golden/config_config4_rank{0,5}.npz (every record of two ranks of eight, from the oracle)"),
        "build_seconds_per_rank": builds,
        "note": "no scaling curve is claimed from this entry at one N: the driver's 1/2/4/8 runs each carry it" if not headline else None,
    }
    if not headline:
        return entry
    # the contract's line for `--scaling strong --workload config4`
    return {"metric": "GACT GCUPS (DP cells/s) on ~10 kb PacBio-shape reads", "value": entry["value"], "unit": "GCUPS", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": entry["ms_per_step"], "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "int16x2 (packed)", "data": "synthetic",
            "config": {"workload": "config4_fixed_job", "tile_size": 320, "tile_overlap": 120, "scoring": "+1/-1/-1/-1",
                       "reads": entry["reads"], "bases": entry["bases"], "candidates": entry["candidates"], "cells_per_step": tot_cells,
                       "parallelism": entry["parallelism"], "slots_in_flight": S, "arch": info["arch"]},
            "config4_strong": entry}


def reference_caller(threads=8):
    """The reference's own UNMODIFIED darwin.cpp (-DGPU, oracle/_ref/darwin_on_hip: built on host/gact.h + gact_shim.cpp) on the
    headline workload's FASTA with eight feeder threads, as a child process behind the timed region: what its threads print
    as "Time GACT calling" (darwin.cpp:408-441: both GACT_Batch calls of a thread), as it is and with GACT_HIP_PAIR_STRANDS=1,
    its lines compared with this repo's own pipeline's (tools/darwin_on_hip_timing.py)."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "darwin_on_hip")
    if not os.path.exists(exe):
        return {"time_gact_calling_ms": None, "note": "oracle/_ref/darwin_on_hip not present on this box (built from /root/reference where that is mounted)"}
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "darwin_on_hip_timing.py"), "ecoli10x", str(threads), "1"],
                           capture_output=True, text=True, timeout=420)
        if p.returncode != 0:
            return {"time_gact_calling_ms": None, "error": (p.stdout + p.stderr)[-400:]}
        d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
        runs = d["runs"]
        return {"caller": "the reference's darwin.cpp -DGPU, unmodified, on host/gact.h + gact_shim.cpp (oracle/_ref/darwin_on_hip)",
                "feeder_threads": d["feeder_threads"], "reads": d["reads"], "candidates": d["candidates"], "cells": d["cells"],
                "time_gact_calling_ms": runs[0]["gact_calling_ms_max_over_threads"], "gcups": runs[0]["gcups_of_the_gact_stage"],
                "launch_ms_on_the_device": sorted(set(c["launch_ms"] for c in runs[0]["shim_split_per_call_us"])),
                "callers_merged": sorted(set(c["merged"] for c in runs[0]["shim_split_per_call_us"])),
                "paired_ms": runs[1]["gact_calling_ms_max_over_threads"] if len(runs) > 1 else None,
                "lines": d["lines"], "lines_equal": bool(d["lines_equal_between_modes_and_to_this_repos_own_pipeline"])}
    except Exception as err:                                    # (reported, never fatal for the line)
        return {"time_gact_calling_ms": None, "error": str(err)[:300]}


def cpp_gather_check(eng, dist, torch, rank, world, n, gathered):
    """One gather of slot 0's records over the C-ABI's RCCL path (include/gact_hip.h gact_hip_comm_create / _gather_lines),
    every rank; rank 0 compares the lines with what the torch.distributed gather of the step delivered.  Run on a thread
    with a time limit: the check must never keep the bench line from being printed.  Whether ANY rank got stuck is
    all-reduced over torch's own communicator, so that every rank takes the same way out (one rank leaving
    through os._exit while the others walk into the final barrier)."""
    global _CPP_GATHER_STUCK
    import tempfile
    import threading
    import uuid
    import numpy as np
    from gact_amd import engine
    tok = [uuid.uuid4().hex if rank == 0 else None]
    dist.broadcast_object_list(tok, src=0)
    path = os.path.join(tempfile.gettempdir(), "gact_rccl_%s.id" % tok[0])
    res = {}

    def work():
        try:
            t0 = time.perf_counter()
            comm = engine.Comm(eng, rank, world, path, timeout_s=30)
            t1 = time.perf_counter()
            counts, lines = comm.gather_lines(n, slot=0)
            t2 = time.perf_counter()
            counts, lines = comm.gather_lines(n, slot=0)              # (the second one: buffers are in place)
            res.update({"create_ms": round((t1 - t0) * 1e3, 1), "first_gather_ms": round((t2 - t1) * 1e3, 2),
                        "gather_ms": round((time.perf_counter() - t2) * 1e3, 2), "counts": [int(c) for c in counts], "lines": lines})
            comm.close()
        except Exception as err:                                       # (reported in the line, not fatal)
            res["error"] = str(err)[:300]

    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(60)
    stuck = torch.tensor([1.0 if th.is_alive() else 0.0], dtype=torch.float64, device=CDEV)
    dist.all_reduce(stuck, op=dist.ReduceOp.MAX)
    if float(stuck.item()) > 0:
        _CPP_GATHER_STUCK = True
        return {"ok": False, "error": "no answer within 60 s on %s" % ("this rank" if th.is_alive() else "another rank")}
    if "error" in res:
        return {"ok": False, "error": res["error"]}
    out = {"ok": True, "records_per_rank": res["counts"], "create_ms": res["create_ms"], "first_gather_ms": res["first_gather_ms"],
           "gather_ms": res["gather_ms"]}
    if rank == 0 and gathered is not None:
        want = np.concatenate([np.ascontiguousarray(g) for g in gathered]) if len(gathered) else np.zeros(0, dtype=engine.Comm.LINE_DTYPE)
        out["equals_torch_distributed_gather"] = bool(want.tobytes() == res["lines"].tobytes())
        out["ok"] = out["equals_torch_distributed_gather"]
    return out


def cpu_baseline(args, cat, offs, rcat, my_cf, rf, my_cr, rr, scoring=(1, -1, -1, -1)):
    """The oracle (CPU restatement, kind "port") timed on this box's host cores on a
    bounded sample of the same candidates; the same sample is the parity gate."""
    import numpy as np
    import oracle_py
    orc = oracle_py.Oracle()
    threads = host_cores()
    # calibrate on a small slice, then size the sample for ~cpu_seconds
    probe = min(len(my_cf), 4 * threads)
    t = time.perf_counter()
    _, cells = orc.gact_many(cat, offs, cat, offs, my_cf[:probe], complement=False, same_file=True, scoring=scoring, n_threads=threads)
    rate = cells / max(time.perf_counter() - t, 1e-6)
    mean_cells = max(cells / max(probe, 1), 1.0)
    n = int(min(len(my_cf), max(probe, args.cpu_seconds * rate / mean_cells)))
    t = time.perf_counter()
    want, cells = orc.gact_many(cat, offs, cat, offs, my_cf[:n], complement=False, same_file=True, scoring=scoring, n_threads=threads)
    dt = time.perf_counter() - t
    fields = ("ref_id", "query_id", "ab", "ae", "bb", "be", "score", "comp", "emitted",
              "first_tile_score", "n_tiles", "cells")
    ok = all(np.array_equal(rf[f][:n], want[f]) for f in fields)
    nr = min(len(my_cr), max(16, n // 8))
    want_r, _ = orc.gact_many(cat, offs, rcat, offs, my_cr[:nr], complement=True, same_file=True, scoring=scoring, n_threads=threads)
    ok = ok and all(np.array_equal(rr[f][:nr], want_r[f]) for f in fields)
    if not ok:
        raise SystemExit("bench.py: PARITY FAILURE between the HIP engine and the oracle on the sample")
    base = {"value": round(cells / dt / 1e9, 4), "unit": "GCUPS", "cores": threads, "kind": "port",
            "sample": "%d forward-strand candidates of this workload (%d cells, %.1f s), oracle/gact_oracle.c, "
                      "%d threads over contiguous candidate ranges" % (n, cells, dt, threads)}
    return base, {"checked_candidates": int(n + nr), "bit_exact": True}


def reference_baseline(reads, my_cf, rf, budget_s=8.0):
    """The reference AS WRITTEN (oracle/_ref/libdarwin_ref.so = the reference's own align.cpp + gact.cpp compiled
    unchanged) on a handful of this workload's candidates, one thread (its GACT keeps state in globals): the figure
    SURVEY.md 8d asks to be quoted next to the port.  Its printed line is compared with the HIP record on the way.
    Skipped with a note where the prebuilt library is absent."""
    import numpy as np
    import oracle_py
    if not oracle_py.ref_available():
        return {"value": None, "kind": "reference", "note": "oracle/_ref/libdarwin_ref.so not present on this box"}
    ref = oracle_py.RefLib()
    orc = oracle_py.Oracle()
    # shortest chains first would flatter nothing: take them in list order, emitted ones, until the budget is spent
    t0 = time.perf_counter()
    cells = n = 0
    for k in range(len(my_cf)):
        if time.perf_counter() - t0 > budget_s:
            break
        c, r = my_cf[k], rf[k]
        if not r["emitted"]:
            continue
        line = ref.gact_line(reads[c["ref_id"]].tobytes(), reads[c["query_id"]].tobytes(), int(c["ref_pos"]),
                             int(c["query_pos"]), ref_id=int(c["ref_id"]), query_id=int(c["query_id"]), ref_name="r", query_name="q")
        if line != orc.format_line(r, "r", "q"):
            raise SystemExit("bench.py: PARITY FAILURE between the HIP engine and the reference's own GACT on candidate %d:\n %s %s"
                             % (k, line, orc.format_line(r, "r", "q")))
        cells += int(r["cells"]); n += 1
    dt = time.perf_counter() - t0
    return {"value": round(cells / dt / 1e9, 5), "unit": "GCUPS", "cores": 1, "kind": "reference",
            "sample": "%d forward-strand candidates of this workload (%d cells, %.1f s) through the reference's own GACT / AlignWithBT "
                      "(align.cpp, gact.cpp compiled unchanged; align.cpp:85 allocates 16.8 MB per tile); lines equal to the HIP records"
                      % (n, cells, dt)}


def feeder_config(workload_name, cat, offs, rcat, cf, cr, want, n_threads=8, steps=6):
    """The headline workload the way the reference drives its GPU (darwin.cpp:408-433,619-629): n_threads feeder threads
    behind a barrier, each with its own engine slot (= GPU_storage) and an n-th of the candidates (dealt round-robin),
    each calling run + fetch `steps` times.  The engine merges runs that arrive together into one launch (the call
    combiner, gact_hip_run_stats.merged_callers).  Records compared with the headline's."""
    import threading
    import numpy as np
    from gact_amd import engine
    eng = engine.Engine(n_slots=n_threads)
    eng.upload(engine.SET_REF, cat, offs); eng.upload(engine.SET_QUERY, cat, offs); eng.upload(engine.SET_QUERY_RC, rcat, offs)
    parts = []
    for k in range(n_threads):
        f, r = cf[k::n_threads], cr[k::n_threads]
        eng.candidates_upload(np.concatenate([f, r]), slot=k)
        out = np.zeros(len(f) + len(r), dtype=engine.OVERLAP_DTYPE)
        eng.register_output(out, slot=k)
        parts.append((len(f), len(r), out))
    gate = threading.Barrier(n_threads + 1)
    errors, merged = [], [0] * n_threads

    def feeder(k):
        try:
            nf, nr, out = parts[k]
            eng.candidates_run_mixed(nf + nr, rc_from=nf, same_file=True, slot=k)       # warm-up, all threads together
            eng.candidates_fetch(nf + nr, slot=k, out=out)
            gate.wait()
            gate.wait()
            for _ in range(steps):
                eng.candidates_run_mixed(nf + nr, rc_from=nf, same_file=True, slot=k)
                eng.candidates_fetch(nf + nr, slot=k, out=out)
            merged[k] = eng.last_run_stats(k)["merged_callers"]
        except Exception as err:
            errors.append(err)
            try:
                gate.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=feeder, args=(k,)) for k in range(n_threads)]
    for t in threads:
        t.start()
    gate.wait()
    t0 = time.perf_counter()
    gate.wait()
    for t in threads:
        t.join()
    dt = (time.perf_counter() - t0) / steps
    if errors:
        raise SystemExit("bench.py: feeder threads failed: %r" % errors[:1])
    nf_all = len(cf)
    for k, (nf, nr, out) in enumerate(parts):
        if out[:nf].tobytes() != want[:nf_all][k::n_threads].tobytes() or out[nf:].tobytes() != want[nf_all:][k::n_threads].tobytes():
            raise SystemExit("bench.py: feeder thread %d's records differ from the headline's" % k)
    cells = int(want["cells"].sum())
    eng.close()
    return {"workload": workload_name + "_self_overlap", "feeder_threads": n_threads, "slots": n_threads, "steps": steps,
            "value": round(cells / dt / 1e9, 2), "unit": "GCUPS", "ms_per_step": round(dt * 1e3, 3),
            "callers_merged_in_last_launch": merged, "records_equal_headline": True}


def side_config(name, args):
    """One of the other single-GPU configurations (BASELINE.json configs[2], configs[4]): built exactly like the
    headline workload, 3 timed steps after 1 warm-up, parity gate against the oracle on a sample."""
    from gact_amd import workload
    blk = workload.make_block(name, candidates=args.candidates)
    cat, offs = blk.rs.concat()
    rcat, roffs = blk.rs.concat(rc=True)
    return timed_config({"workload": name + "_self_overlap"}, cat, offs, rcat, roffs, blk.cf, blk.cr)


# the headline workload again under the conditions that take it off its fastest kernels (each line says which kernels ran):
# reads with N runs and soft-masked stretches (align.cpp:134 compares raw bytes: those candidates leave the 2-bit image),
# the affine pass on the same linear scoring, a truly affine scoring, the int32 kernel
SIDE_SLOTS = 4          # steps in flight in other_configs / variants (set from --slots)


VARIANTS = (
    {"label": "1% of the reads hold an N run and a lower-case stretch", "dirty_fraction": 0.01},
    {"label": "affine-gap pass forced on the linear scoring (GACT_HIP_NO_LIN=1)", "env": {"GACT_HIP_NO_LIN": "1"}},
    {"label": "affine scoring match 2, mismatch -3, gap open -5, gap extend -2", "scoring": (2, -3, -5, -2)},
    {"label": "int32 kernel forced (GACT_HIP_FORCE_INT32=1)", "env": {"GACT_HIP_FORCE_INT32": "1"}},
)


def variant_config(v, workload_name, reads, cf, cr):
    import numpy as np
    from gact_amd import synth
    if v.get("dirty_fraction"):
        rng = np.random.default_rng(20260903)
        reads = list(reads)
        picks = rng.choice(len(reads), max(1, int(round(len(reads) * v["dirty_fraction"]))), replace=False)
        for k in picks:
            r = reads[k].copy()
            if len(r) > 600:
                a = int(rng.integers(100, len(r) - 400))
                r[a:a + 25] = ord("N")
                b = int(rng.integers(100, len(r) - 400))
                r[b:b + 60] = np.frombuffer(bytes(r[b:b + 60]).lower(), dtype=np.uint8)
            reads[k] = r
    offs = np.zeros(len(reads) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    cat = np.concatenate(reads)
    rcat = np.concatenate([synth.revcomp(r) for r in reads])
    saved = {k: os.environ.get(k) for k in v.get("env", {})}
    os.environ.update(v.get("env", {}))
    try:
        return timed_config({"workload": workload_name + "_self_overlap", "variant": v["label"], "variant_index": VARIANTS.index(v)},
                            cat, offs, rcat, offs, cf, cr, scoring=v.get("scoring", (1, -1, -1, -1)))
    finally:
        for k, old in saved.items():
            if old is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = old


def timed_config(head, cat, offs, rcat, roffs, cf, cr, scoring=(1, -1, -1, -1)):
    import numpy as np
    import oracle_py
    from gact_amd import engine
    S = SIDE_SLOTS
    eng = engine.Engine(n_slots=S, scoring=scoring)
    info, eng_tile, eng_overlap = eng.device_info(), eng.tile_size, eng.tile_overlap
    eng.upload(engine.SET_REF, cat, offs); eng.upload(engine.SET_QUERY, cat, offs); eng.upload(engine.SET_QUERY_RC, rcat, roffs)
    nf, nr = len(cf), len(cr)
    cands = np.concatenate([cf, cr])
    recs = []
    for k in range(S):
        eng.candidates_upload(cands, slot=k)
        recs.append(np.zeros(nf + nr, dtype=engine.OVERLAP_DTYPE))
        eng.register_output(recs[k], slot=k)
    rec = recs[0]
    # one step at a time on slot 0 (1 warm-up + 3 timed): the kernels' own times
    steps, stats = 3, []
    for k in range(1 + steps):
        if k == 1:
            eng.sync(0)
            t0 = time.perf_counter()
        eng.candidates_run_mixed(nf + nr, rc_from=nf, same_file=True)
        eng.candidates_fetch(nf + nr, out=rec)
        if k:
            stats.append(eng.last_run_stats())
    dt_single = (time.perf_counter() - t0) / steps
    # S steps in flight, like the headline (S untimed, 2 S timed)
    dt = dt_single
    flight = stats[-1]
    if S > 1:
        def run_steps(n):
            for k in range(n):
                eng.candidates_run_mixed(nf + nr, rc_from=nf, same_file=True, slot=k % S)
                if k >= S - 1:
                    eng.candidates_fetch(nf + nr, slot=(k - (S - 1)) % S, out=recs[(k - (S - 1)) % S])
            for k in range(max(n - (S - 1), 0), n):
                eng.candidates_fetch(nf + nr, slot=k % S, out=recs[k % S])
        eng.set_option("runs_in_flight", 1)          # (every launch the throughput layout, the first ones too: see main())
        run_steps(S)
        t0 = time.perf_counter()
        run_steps(2 * S)
        dt = (time.perf_counter() - t0) / (2 * S)
        eng.set_option("runs_in_flight", 0)
        for k in range(1, S):
            if recs[k].tobytes() != rec.tobytes():
                raise SystemExit("bench.py: slot %d's records differ from slot 0's on %s" % (k, head))
        steps = 2 * S
        flight = eng.last_run_stats(S - 1)
    cells = int(rec["cells"].sum())
    st = stats[-1]
    # parity gate: a strided sample of both strands through the oracle, ~4 s of host time; every candidate the engine
    # aligned from raw bytes beside the 2-bit launches is in it when there are few
    orc = oracle_py.Oracle()
    threads = host_cores()
    fields = ("ref_id", "query_id", "ab", "ae", "bb", "be", "score", "comp", "emitted", "first_tile_score", "n_tiles", "cells")
    target = 1.0e10                                  # cells through the oracle
    stride = max(1, int(cells / target))
    checked = 0
    for comp, sl, qcat, qoffs in ((False, slice(0, nf), cat, offs), (True, slice(nf, nf + nr), rcat, roffs)):
        c, got = cands[sl][::stride], rec[sl][::stride]
        want, _ = orc.gact_many(cat, offs, qcat, qoffs, c, complement=comp, same_file=True, scoring=scoring, n_threads=threads)
        if not all(np.array_equal(got[f], want[f]) for f in fields):
            raise SystemExit("bench.py: PARITY FAILURE between the HIP engine and the oracle on %s" % head)
        checked += len(c)
    eng.close()
    out = dict(head)
    out.update({"value": round(cells / dt / 1e9, 2), "unit": "GCUPS", "steps": steps, "warmup": S if S > 1 else 1,
                "ms_per_step": round(dt * 1e3, 3), "slots_in_flight": S,
                "single_slot": {"value": round(cells / dt_single / 1e9, 2), "ms_per_step": round(dt_single * 1e3, 3), "steps": 3}, "scoring": "%+d/%+d/%+d/%+d" % tuple(scoring), "candidates": int(nf + nr),
                "raw_byte_candidates": int(st["raw_candidates"]),
                "tiles": int(rec["n_tiles"].sum()), "cells_per_step": cells,
                # (a launch made while another slot is running takes the layout with the better throughput)
                "kernel_layout": flight["layout"] + ("-lin" if flight["linear_gap"] else "-aff" if flight.get("affine_drift") else "") + ("-coop" if flight.get("coop_walks") else ""),
                "single_slot_kernel_layout": st["layout"] + ("-lin" if st["linear_gap"] else "-aff" if st.get("affine_drift") else "") + ("-coop" if st.get("coop_walks") else ""),
                "kernel_ms": round(float(np.mean([x["main_ms"] for x in stats])), 3),
                "seed_kernel_ms": round(float(np.mean([x["seed_ms"] for x in stats])), 3),
                "parity": {"checked_candidates": int(checked), "bit_exact": True}})
    if st["packed16"]:
        # the main launch of the one-at-a-time leg against the issue peak (counters: the committed PMC pass of this workload)
        wl = head["workload"].replace("_self_overlap", "") + ("" if head.get("variant") is None else "_variant_%d" % head.get("variant_index", 0))
        out["roofline"] = short_roofline(st, float(np.mean([x["main_ms"] for x in stats])), cells - int(st["seed_cells"]), info,
                                         wl, eng_tile, eng_tile - eng_overlap)
        out["roofline"]["seed"] = seed_roofline(st, float(np.mean([x["seed_ms"] for x in stats])), int(st["seed_cells"]), info, None) if st["seed_cells"] else None
    return out


if __name__ == "__main__":
    main()
