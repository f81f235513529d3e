"""bench.py — Quick-ADC flat scan on MI355X: PQ codes scanned / second (+ Recall@100).

Workload (BASELINE.json configs[3] shape, which fits one GPU): flat database of 1B synthetic
16x4 PQ codes (8 B/code, counter-based generator), R = 100, keep = 1 %, every query scans the
whole list, through the whole scanner_4::query_scan path (float pre-scan of the starts -> qmax,
quantizer, int8 scan of every code, candidate replay).

ONE JSON line, several legs (SURVEY.md §8d keeps the two scan modes apart):

* headline `value` / `ms_per_step` / `roofline` (the timed region of the contract, ONE region for all three): the METRIC'S
  MODE — ONE QUERY PER PASS over the list, what the reference's scan_avx_4 does (simd_scan.hpp:125-187, one call per query at
  db_query_4.cpp:287-308).  A step = one batch of NQ = 32 queries (the reference's documented `-b32`, README.md:275-330), each
  of which walks the whole list BY ITSELF (library options share_variant = 0, mq = 0: no sibling launch, no multi-query
  pass; scan_i8_kernel<M,2,nt,chunk>): 8 B per (code, query) cross the HBM interface — `roofline.traffic` (rocprofv3 PMC
  passes over the same kind of steps, collected IN THIS RUN by child processes, --pmc-leg; FETCH_SIZE x2 on gfx950, WRITE_SIZE,
  separate passes) is within 2 % of the algorithmic bytes.  `roofline` (bound "hbm", frac <= 1) = algorithmic bytes of the
  region's streaming launches / their HIP-event time on the library's stream.  With --gpus N the SAME mode runs on every rank
  (its shard of the list, one gather per 32-query step) and `roofline` is rank 0's shard (traffic null: no PMC child at N > 1).
* `value_batched` / `ms_per_step_batched` / `roofline_batched` (reported SEPARATELY): the same 32-query
  steps with the queries launched as L2-sharing siblings, 8 queries per pass (scan_i8_mq_kernel) — the codes cross the HBM
  interface about once per LAUNCH, the kernel is LDS-bound (bound "lds"); timed in its own region after the headline's.
* `roofline_one_query_per_call`: the strictest reading of the reference's mode — sequential single-query batches (three in
  flight), 7 bound-level launches per query — in its own short region; `value_one_query_per_call` is its wall-clock rate.
* `roofline_32x4`: the same one-query-per-pass leg on 1B x 32x4 codes (16 B/code), with its own in-run PMC traffic.
* `roofline_ivf` / `roofline_ivf_c5`: the launches of the IVF legs' partition-major second phase against their roofs
  (grouped scan: LDS lookup rate; head: HBM), from a short profiled pass after the timed loops.
* `ivf`: BASELINE configs[2] shape (100M codes in K=4096 labelled-free synthetic partitions, nprobe 32,
  1024-query pipelined batches through the device-side feeders).
* `ivf_c5_one_gpu`: the same leg at BASELINE configs[4]'s shape on one GPU (1B x 32x4 codes, K=16384, 96-d, nprobe 64).
* `c2` / `roofline_c2`: BASELINE configs[1] (flat 10M x 16x4), both modes: 32-query steps and one query per pass.
* `latency_us_single_query`: synchronous single query on a 10^5-code list (README.md:327-330: 86 us);
  `c2.one_query_synchronous`: the same on BASELINE configs[1]'s 10^7-code list.
* `recall_at_100_real_encode` (flat, 10M) / `recall_at_100_real_encode_ivf` (K = 4096, nprobe 32): recall on REAL encodings against the
  exact float L2 nearest neighbour, with `reference_heaps_equal` (the reference's scan_avx_4 on the same codes and int8 tables).
* `cpu_baseline`: the reference's own scan_avx_4<16> (oracle/_ref), 1 thread; `cpu_baseline_all_cores`:
  the same kernel on every physical core (C++ threads inside oracle/_ref, pinned, per-thread copies);
  `cpu_baseline_32x4`: scan_avx_4<32> on the 32x4 list; `cpu_baseline_ivf` / `cpu_baseline_ivf_c5`: the reference's
  scan over the probed partitions of the IVF legs' own first 32 queries with the device's int8 tables (1 thread each).

With --gpus N and no WORLD_SIZE in the environment the script launches its own N ranks
(torch.distributed.run, one per GPU, RCCL) BEFORE touching the GPU and relays rank 0's line; under
the driver's torchrun launch it is a rank.  The SAME 1B-code list is sharded over the N ranks in
contiguous ranges ("strong" scaling; keys are 32-bit as in the reference, so the list cannot grow
past 2^32 anyway) and the per-shard push streams are gathered once per batch and replayed.

    python bench.py --gpus 1 --steps 20 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Environment overrides for quick runs: QADC_BENCH_CODES, QADC_BENCH_NQ, QADC_BENCH_M,
QADC_BENCH_CPU_SECONDS (0 disables the CPU legs), QADC_BENCH_SINGLE_QUERIES (0 = no one-query-per-call leg), QADC_BENCH_BATCHED
(0 = no batched-mode leg), QADC_BENCH_PMC (0 = no
in-run PMC child passes), QADC_BENCH_IVF_CODES (0 = no IVF leg), QADC_BENCH_IVF_C5 (0 = no 1B x 32x4 IVF leg), QADC_BENCH_32X4 (0 = no 32x4 leg),
QADC_BENCH_REAL_CODES (0 = no real-encode recall leg), QADC_BENCH_LATENCY (0 = no latency leg), QADC_BENCH_CEILING (0 = no
measured-ceiling leg).
"""
import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "quick-adc_amd"))
# Kernel arguments in device memory (a HIP runtime setting, read when the runtime starts): a query kernel's first dependent reads —
# its partition descriptors, and a lone small query's whole input, which rides in the kernel-argument segment — then come from HBM
# instead of host memory over PCIe.  Synchronous single query: 54.9 -> 51.5 us on the same box (profiles/r06_latency_sweep.txt;
# the kernel-side work of round 6 took it on to 32 us: profiles/r06_lone_query_latency_ab.txt);
# nothing else moves.  A serving process sets the same variable.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
LDS_PEAK_GCYC = 256 * 2.4  # LDS-array cycles per second: 256 CUs x 2.4 GHz nominal (same guide)
R, KEEP, SEED = 100, 0.01, 0x5EED0001
# the two scan modes of a 32-query step (library options):
#   one query per pass (the metric's mode, the headline): no sibling launch, no multi-query pass — every query of the step walks
#     the list by itself; front_run_max = 0 keeps every streaming launch on the main stream inside the HIP-event-timed groups
#   batched: the library's defaults — queries of a step as L2-sharing siblings, 8 per pass
MODE_ONE_QUERY_PER_PASS = dict(share_variant=0, mq=0, front_run_max=0)
MODE_BATCHED = dict(share_variant=0x41, mq=1, front_run_max=2 << 20)


def set_mode(idx, mode):
    for k, v in mode.items():
        idx.set_option(k, v)


def make_tables(rng, codebooks, nq):
    """Float distance tables ||q_m - c||^2 of N(0,1) queries against N(0,1) codebooks [M][16][d]."""
    M, _, d = codebooks.shape
    q = rng.normal(size=(nq, M, 1, d)).astype(np.float32)
    t = ((q - codebooks[None]) ** 2).sum(-1, dtype=np.float32)
    return np.ascontiguousarray(t.reshape(nq, 1, M * 16), np.float32)


# --------------------------------------------------------------------------------------------- CPU legs
def cpu_baseline(M, n_total, qtables, seconds):
    """Times the reference's own scan_avx_4<M> (oracle/_ref, built from /root/reference) — or, if
    that build is absent, the oracle's scalar C port — on a bounded prefix of the same synthetic
    list with the same int8 tables, one thread.  Reported, never the target."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    cs = M // 2
    n = int(min(n_total, 32 * 1024 * 1024))
    codes = po.fill_codes(0, (n * cs + 7) // 8, SEED)[:n * cs].reshape(n, cs)
    kind = "reference" if po.have_ref() else "port"
    if kind == "reference":
        inter = po.ref_interleave(codes)
        run = lambda qt: po.ref_scan_interleaved(M, [inter], [n], None, qt, R)
    else:
        run = lambda qt: po.scan_i8(M, [codes], None, qt, R)
    run(qtables[0])  # warm
    t0, nqueries = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        run(qtables[nqueries % len(qtables)])
        nqueries += 1
    dt = time.perf_counter() - t0
    model, cpus = po.host_topology()
    return {"value": n * nqueries / dt, "unit": "codes/s", "cores": 1, "kind": kind,
            "cpu_model": model, "physical_cores_usable": len(cpus),
            "build_flags": "g++ -std=c++14 -O3 -m64 -mavx2 -mfma -mpopcnt -mbmi2 -ffast-math (the reference's "
                           "CMakeLists.txt:7 flags with -march=native replaced by that explicit ISA set so that the "
                           "prebuilt library runs on any AVX2 host)" if kind == "reference" else "gcc -std=c11 -O2",
            "sample": "%d queries x first %d codes of the same synthetic list, same int8 tables, R=%d, "
                      "1 thread, %s" % (nqueries, n, R, "scan_avx_4<%d> compiled from the reference" % M
                                        if kind == "reference" else "scalar C port (oracle)")}


def cpu_ivf_baseline(sample, seconds):
    """The reference's CPU path beside an IVF leg: scan_avx_4<M> through qadc_ref_scan (oracle/_ref:
    (0,127) sentinel, then every probed partition in assign[] order into ONE heap, db_query_4.cpp:276, 287-308) on the probed
    partitions of the leg's own first queries — regenerated on the CPU from the partitions' generator streams, laid out by the
    reference's interleave_partition_4 — with the DEVICE's int8 tables (qadc_slot_qtables), 1 thread.  The heaps are compared
    with the GPU's while at it.  Reported, never the target."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    from concurrent.futures import ThreadPoolExecutor
    if not po.have_ref():
        return {"error": "oracle/_ref not built"}
    M, sizes, seed0 = sample["M"], sample["sizes"], sample["seed0"]
    assign, qt = sample["assign"], sample["qtables"]
    cs, nq = M // 2, assign.shape[0]
    need = sorted(set(int(p) for p in assign.reshape(-1)))

    def make(p):                                               # partition p: generator stream seed0 + p, words from 0 (ivf_leg)
        n = int(sizes[p])
        if n == 0:
            return p, np.zeros(0, np.uint8)
        codes = po.fill_codes(0, (n * cs + 7) // 8, seed0 + p)[:n * cs].reshape(n, cs)
        return p, po.ref_interleave(codes)
    model, cpus = po.host_topology()
    t_gen = time.perf_counter()
    with ThreadPoolExecutor(max(1, min(len(cpus), 16))) as ex:     # (data generation only; the timed scan below is one thread)
        inter = dict(ex.map(make, need))
    t_gen = time.perf_counter() - t_gen

    def run(q):
        parts = [inter[int(p)] for p in assign[q]]
        return po.ref_scan_interleaved(M, parts, [int(sizes[int(p)]) for p in assign[q]], None, qt[q], R)
    same = 0
    for q in range(nq):
        k, v = run(q)
        n = int(sample["heap_sizes"][q])
        same += int(np.array_equal(k, sample["keys"][q, :n]) and np.array_equal(v, sample["values"][q, :n]))
    t0, done, codes_done = time.perf_counter(), 0, 0
    while time.perf_counter() - t0 < seconds:
        run(done % nq)
        codes_done += int(sizes[assign[done % nq]].sum())
        done += 1
    dt = time.perf_counter() - t0
    return {"value": codes_done / dt, "unit": "codes/s", "cores": 1, "kind": "reference", "cpu_model": model,
            "us_per_query": dt * 1e6 / done, "reference_heaps_equal": "%d/%d" % (same, nq),
            "sample": "%d query scans in %.1f s cycling over the first %d queries of the leg's first batch: their %d probed partitions each "
                      "(%d distinct partitions regenerated on the CPU in %.1f s, untimed), the device's int8 tables, one shared heap per query, "
                      "scan_avx_4<%d> compiled from the reference, 1 thread" % (done, dt, nq, assign.shape[1], len(need), t_gen, M)}


def cpu_extra_legs(M, n_total, qtables, seconds):
    """Two more CPU figures asked for by BASELINE.md §3 (reported, never the target):
    - the reference's AVX2 scan on ALL physical host cores, driven from C++ inside oracle/_ref (one pinned thread
      per physical core, each scanning its own first-touched copy of the sample, whole queries back to back);
    - BASELINE config 1, PQ 8x8 float ADC over 1M codes (scanner_simple / scan_standard<uint8_t,8>), 1 thread,
      timed with the oracle's C port."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    out = {}
    if po.have_ref():
        cs = M // 2
        model, cpus = po.host_topology()
        # per-thread sample: 8 Mi codes (64 MB for 16x4) — larger than a core's share of the caches, so every
        # thread streams from DRAM like a 1B-code scan would; bounded so that all copies stay below ~16 GB
        n = int(min(n_total, 8 * 1024 * 1024, (16 << 30) // (len(cpus) * cs)))
        codes = po.fill_codes(0, (n * cs + 7) // 8, SEED)[:n * cs].reshape(n, cs)
        inter = po.ref_interleave(codes)
        done, dt = po.ref_scan_mt(M, inter, n, qtables, R, cpus, seconds)
        out["cpu_baseline_all_cores"] = {
            "value": n * done / dt, "unit": "codes/s", "cores": len(cpus), "kind": "reference", "cpu_model": model,
            "sample": "%d whole queries x %d codes in %.1f s: one pinned C++ thread per physical core this process may "
                      "keep busy (%d: physical cores in its affinity mask, capped by the container's cgroup CPU quota), "
                      "each with its own first-touched copy of the first %d codes of the list" % (done, n, dt, len(cpus), n)}
    rng = np.random.default_rng(5)
    codes8 = rng.integers(0, 256, (1000000, 8), dtype=np.uint8)
    tables8 = rng.random((1, 8, 256)).astype(np.float32)
    po.scan_standard_u8(8, [codes8], None, tables8, R)
    t0, done = time.perf_counter(), 0
    while time.perf_counter() - t0 < min(seconds, 3.0):
        po.scan_standard_u8(8, [codes8], None, tables8, R)
        done += 1
    dt = time.perf_counter() - t0
    out["cpu_config1_pq8x8_float_adc"] = {"value": 1000000 * done / dt, "unit": "codes/s", "cores": 1, "kind": "port",
                                          "sample": "%d queries x 1M codes, scan_standard<uint8_t,8> C port" % done}
    return out


# --------------------------------------------------------------------------------------------- recall on real encodings
def real_encode_recall(M, n, nq, local_rank):
    """Recall@R on REAL encodings (SURVEY.md §8d "real-encode variant"): clustered synthetic 128-d vectors,
    codebooks = sampled sub-vectors, PQ-encoded on the GPU (qadc_pq_encode), queried through the device-side
    feeders (qadc_search); ground truth = exact float L2 nearest neighbour (torch, chunked).  Untimed.
    The same queries also go through the host-table entry point and the resulting int8 tables are handed to the
    reference's own scan_avx_4 (oracle/_ref) over the same codes: `reference_heaps_equal` says whether the
    reference kernel ends with the same heaps — i.e. whether this recall IS the reference path's recall."""
    import torch
    import pyqadc
    dev = torch.device("cuda", local_rank)
    g = torch.Generator(device=dev).manual_seed(7)
    dim, ds, cs = 128, 128 // M, M // 2
    C = max(2000, n // 300)     # ~300 points per cluster: neither trivial nor hopeless for a 100-entry shortlist
    centres = torch.randn(C, dim, device=dev, generator=g) * 3
    base = centres[torch.randint(0, C, (n,), device=dev, generator=g)] + torch.randn(n, dim, device=dev, generator=g)
    queries = centres[torch.randint(0, C, (nq,), device=dev, generator=g)] + torch.randn(nq, dim, device=dev, generator=g)
    ids = torch.randint(0, n, (M, 16), device=dev, generator=g)
    cb = torch.stack([base[ids[m], m * ds:(m + 1) * ds] for m in range(M)]).contiguous()      # [M][16][ds]
    raw = torch.zeros(n * cs + 64, dtype=torch.uint8, device=dev)                             # padded tail
    torch.cuda.synchronize()
    cb_host = cb.cpu().numpy()
    pyqadc.pq_encode_device(cb_host, base.data_ptr(), n, dim, raw.data_ptr(), local_rank)
    best_d = torch.full((nq,), float("inf"), device=dev)
    best_i = torch.zeros(nq, dtype=torch.long, device=dev)
    for lo in range(0, n, 1 << 20):
        d = torch.cdist(queries, base[lo:lo + (1 << 20)])
        dmin, imin = d.min(dim=1)
        upd = dmin < best_d
        best_d = torch.where(upd, dmin, best_d)
        best_i = torch.where(upd, imin + lo, best_i)
    gt = best_i.cpu().numpy()
    torch.cuda.synchronize()
    idx = pyqadc.Index(M, local_rank)
    idx.add_partition_device(raw.data_ptr(), n, keepalive=raw)
    idx.finalize(KEEP)
    idx.set_pq(cb_host)
    qh = queries.cpu().numpy()
    res = idx.search(qh, 1, R)
    hits = sum(int(gt[q] in set(res["keys"][q][:res["sizes"][q]].tolist())) for q in range(nq))
    out = {"value": hits / nq, "codes": n, "queries": nq,
           "data": "synthetic 128-d vectors, %d clusters (3*N(0,1) centres + N(0,1)), codebooks = sampled sub-vectors, PQ %dx4 "
                   "encoded on the GPU; ground truth = exact float L2 NN" % (C, M)}
    # the reference kernel on the same codes with the same int8 tables (CPU, bounded: n codes x nq queries)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    if po.have_ref() and float(os.environ.get("QADC_BENCH_CPU_SECONDS", 15)) > 0:
        tables = np.ascontiguousarray(((qh.reshape(nq, M, 1, ds) - cb_host[None]) ** 2).sum(-1, dtype=np.float32)
                                      .reshape(nq, 1, M * 16), np.float32)
        r2 = idx.query_scan(np.zeros((nq, 1), np.int32), tables, R, want_qtables=True)
        codes_h = raw[:n * cs].cpu().numpy().reshape(n, cs)
        inter = po.ref_interleave(codes_h)
        same, hits_ref = 0, 0
        for q in range(nq):
            k, v = po.ref_scan_interleaved(M, [inter], [n], None, r2["qtables"][q], R)
            same += int(np.array_equal(k, r2["heaps"][q][0]) and np.array_equal(v, r2["heaps"][q][1]))
            hits_ref += int(gt[q] in set(k.tolist()))
        out["reference_heaps_equal"] = "%d/%d" % (same, nq)
        out["reference_recall_at_100"] = hits_ref / nq
        out["reference_note"] = ("the reference's scan_avx_4<%d> (oracle/_ref) run on the same codes with this engine's int8 "
                                 "tables of the host-table entry point; heaps compared array for array" % M)
    idx.close()
    return out


# --------------------------------------------------------------------------------------------- HBM traffic (PMC)
def under_profiler():
    return any(("rocprof" in (os.environ.get(k) or "").lower()) for k in ("LD_PRELOAD", "HSA_TOOLS_LIB", "ROCP_TOOL_LIB")) \
        or any(k.startswith("ROCPROF") for k in os.environ)


def pmc_leg_main():
    """Child process of the in-run PMC passes (runs under `rocprofv3 --pmc ...`): two steps of the headline's mode (NQ queries per
    step, one query per pass), no torch.  Prints the profile counters it needs as one JSON line."""
    import pyqadc
    M = int(os.environ.get("QADC_BENCH_M", 16))
    N = int(float(os.environ.get("QADC_BENCH_CODES", 1e9)))
    NQ = int(os.environ.get("QADC_BENCH_NQ", 32))
    idx = pyqadc.Index(M, 0)
    idx.add_partition_synthetic_shard(N, 0, N, SEED, max(1, int(np.float32(N) * np.float32(KEEP))))
    idx.finalize(KEEP)
    idx.set_option("profile", 1)
    set_mode(idx, MODE_ONE_QUERY_PER_PASS)
    for kv in filter(None, os.environ.get("QADC_BENCH_OPTS", "").split(",")):     # tuning experiments only (as in main)
        idx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
    rng = np.random.default_rng(1234)
    codebooks = rng.normal(size=(M, 16, 128 // M)).astype(np.float32)
    tb = make_tables(rng, codebooks, NQ)
    assign = np.zeros((NQ, 1), np.int32)
    for _ in range(2):
        idx.query_scan(assign, tb.copy(), R)
    p = idx.profile()
    print(json.dumps({"pmc_leg": True, "scan_launches": p["scan_launches"], "scan_codes": p["scan_codes"]}), flush=True)
    idx.close()


def pmc_traffic_in_run(M, N, nq_step):
    """Runs two steps of the headline's mode (pmc_leg_main) twice under rocprofv3 (FETCH_SIZE, then WRITE_SIZE: they do not fit one pass,
    MI355X_MICROARCH.md "rocprofv3 PMC slots") and returns HBM bytes per scan_i8_kernel launch, or (None, reason).
    FETCH_SIZE is in KB and counts half of a 16-B/lane streaming read on gfx950 (same guide, "HBM"): x 1024 x 2."""
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    if under_profiler():
        return None, "bench.py itself runs under a profiler"
    tmp = tempfile.mkdtemp(prefix="qadc_pmc_")
    env = dict(os.environ, TMPDIR="/tmp", QADC_BENCH_M=str(M), QADC_BENCH_CODES=str(N), QADC_BENCH_NQ=str(nq_step))
    got = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.abspath(__file__), "--pmc-leg"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
            for line in r.stdout.decode(errors="replace").splitlines():      # the leg's own launch / code counters
                if line.startswith("{") and "pmc_leg" in line:
                    got["leg"] = json.loads(line)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s failed (rc %d): %s" % (counter, r.returncode, r.stderr.decode()[-300:])
            per = {}
            for row in csv.DictReader(open(files[0])):
                name = row["Kernel_Name"]
                if "scan_i8_kernel" in name and row["Counter_Name"] == counter:
                    per[row["Dispatch_Id"]] = per.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
            if not per:
                return None, "no scan_i8_kernel dispatch in the %s pass" % counter
            got[counter] = (sum(per.values()), len(per))
    except Exception as e:  # noqa: BLE001 — the PMC leg is optional evidence, never fatal
        return None, "in-run PMC pass failed: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    (fk, fl), (wk, wl) = got["FETCH_SIZE"], got["WRITE_SIZE"]
    leg = got.get("leg")
    if not leg or leg.get("scan_launches") != fl or not leg.get("scan_codes"):
        return None, "the PMC leg's launch count (%s) does not match the counter file's (%d)" % (leg and leg.get("scan_launches"), fl)
    # per launch like `achieved`; the leg runs 2 steps, the timed region K: the comparable figure is bytes over the
    # ALGORITHMIC bytes of the same launches (same launch shapes in both)
    total = fk * 1024 * 2 + wk * 1024
    alg_leg = float(leg["scan_codes"]) * (M // 2)
    return {"bytes_per_launch_of_the_pmc_leg": total / fl, "launches": fl, "fetch_kb_total": fk, "write_kb_total": wk,
            "algorithmic_bytes_of_those_launches": alg_leg, "traffic_over_algorithmic": total / alg_leg}, "in-run rocprofv3 --pmc passes"


def pmc_traffic_from_profiles(M, N, alg_bytes_per_launch):
    """Fallback: the committed one-query-per-pass PMC profile, only if it was taken on exactly this mode/config."""
    f = os.path.join(ROOT, "profiles", "r06_headline_hbm_traffic.json")
    if not os.path.exists(f):
        return None, "no in-run PMC pass and no committed profile"
    pj = json.load(open(f))
    if pj.get("mode") != "one_query_per_pass_batch32" or pj.get("codes") != N or pj.get("M") != M:
        return None, "committed PMC profile is for another mode/config (%s, %s codes, M=%s)" % (
            pj.get("mode"), pj.get("codes"), pj.get("M"))
    return pj["traffic_over_algorithmic"] * alg_bytes_per_launch, "profiles/r06_headline_hbm_traffic.json (ratio x this run's bytes)"


# --------------------------------------------------------------------------------------------- GPU side legs
def single_query_leg(idx, M, N, pool, nqueries, depth=3):
    """ONE query per pass (the reference's mode): `nqueries` sequential single-query batches, `depth` (three) in flight.
    Returns the HIP-event profile of the streaming launches and the wall-clock rate."""
    import torch
    a1 = np.zeros((1, 1), np.int32)
    tabs = [pool[i % len(pool)][j:j + 1].copy() for i in range(2) for j in range(pool[0].shape[0])]

    def run(k):
        pend = []
        for s in range(k):
            idx.submit(s % depth, a1, tabs[s % len(tabs)].copy(), R)
            pend.append(s % depth)
            if len(pend) == depth:
                idx.collect(pend.pop(0))
        while pend:
            idx.collect(pend.pop(0))

    run(2 * depth)
    idx.profile_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(nqueries)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return idx.profile(), dt


def one_pass_batch_leg(idx, pool, steps, warm=2, depth=3):
    """The headline's mode on another list: `steps` pipelined steps of len(pool[0]) queries, every query walking the list by
    itself (MODE_ONE_QUERY_PER_PASS).  Returns the HIP-event profile of the streaming launches and the wall time."""
    import torch
    nq = pool[0].shape[0]
    assign = np.zeros((nq, 1), np.int32)
    set_mode(idx, MODE_ONE_QUERY_PER_PASS)

    def run(k):
        pend = []
        for s in range(k):
            idx.submit(s % depth, assign, pool[s % len(pool)].copy(), R)
            pend.append(s % depth)
            if len(pend) == depth:
                idx.collect(pend.pop(0))
        while pend:
            idx.collect(pend.pop(0))

    run(warm)
    idx.profile_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    return idx.profile(), time.perf_counter() - t0


def roofline_single(M, N, prof, dt, nqueries, traffic, traffic_src, region=None):
    """HBM roofline object of a one-query-per-pass region from the library's HIP-event profile of its streaming launches.
    region = None: sequential single-query batches (one query per CALL); else the text that describes the timed region."""
    cs = M // 2
    scan_ms = prof["scan_ms"]
    launches = max(prof["scan_launches"], 1)
    alg = prof["scan_codes"] * cs / launches
    achieved = prof["scan_codes"] * cs / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_src,
            "traffic_over_algorithmic": None if traffic is None else traffic / alg,
            "kernel": "scan_i8_kernel<%d,2,nt,chunk>" % M,
            "mode": "one query per pass over the whole list (simd_scan.hpp:125-187 called once per query, "
                    "db_query_4.cpp:287-308): every query streams the list from HBM by itself",
            "timed_region": region or ("%d sequential single-query batches (three in flight), HIP events around every run of "
                                       "consecutive streaming launches on the library's stream" % nqueries),
            "launches": prof["scan_launches"], "avg_launch_ms": scan_ms / launches,
            "algorithmic_bytes_per_launch": alg,
            "algorithmic_bytes_rule": "%d B per (code, query) (SURVEY.md 8d) x codes of the launch's bound level" % cs,
            "codes_per_sec_wall": float(N) * nqueries / dt, "ms_per_query_wall": dt * 1e3 / nqueries,
            "region_wall_s": dt, "region_queries": nqueries,
            "small_run_codes_not_event_timed": prof["small_codes"]}


def ivf_leg(local_rank, M=16, K=4096, MA=32, dim=128, N=None, seed0=1000, shard=None):
    """BASELINE configs[2] shape: 100M x 16x4 codes in K = 4096 ragged partitions, nprobe 32, R = 100, queries in
    (coarse assignment, residual tables, pre-scan, quantizer, scan, heap all on the GPU), 1024-query batches pipelined.
    shard = None: the whole database on this GPU, qadc_search_submit / _collect.
    shard = dict(rank, world, placement, init): this rank's part of it — placement "whole" = whole partitions per rank,
    size-balanced (qadc_place_partitions), "range" = every partition range-split over the ranks; every rank holds the
    starts of every partition (the pre-scan and hence qmax / the int8 tables need no exchange), submits the same query
    batches, and qadc_dist_collect gathers the push streams (init(idx) sets the merge up: RCCL, or the loopback stand-in
    of tools/ivf_shard_sizes.py)."""
    import pyqadc
    NQB = 1024
    if N is None:
        N = int(float(os.environ.get("QADC_BENCH_IVF_CODES", 1e8)))
    rng = np.random.default_rng(0)
    sizes = rng.multinomial(N, np.ones(K) / K)
    idx = pyqadc.Index(M, local_rank)
    local_codes = 0
    if shard is None:
        for p in range(K):
            idx.add_partition_synthetic(int(sizes[p]), seed0 + p)
        local_codes = int(sizes.sum())
    else:
        rank, world = shard["rank"], shard["world"]
        owner = pyqadc.place_partitions(sizes, world) if shard["placement"] == "whole" else None
        for p in range(K):
            n = int(sizes[p])
            if n == 0:
                idx.add_partition_synthetic(0, seed0 + p)
                continue
            if owner is not None:
                first, ln = (0, n) if owner[p] == rank else (0, 0)
            else:
                per = (n // world) // 16 * 16
                first, ln = rank * per, (n - rank * per) if rank == world - 1 else per
            idx.add_partition_synthetic_shard(n, first, ln, seed0 + p, max(1, int(np.float32(n) * np.float32(KEEP))))
            local_codes += ln
    idx.finalize(KEEP)
    cb = rng.normal(size=(M, 16, dim // M)).astype(np.float32)
    coarse = rng.normal(size=(K, dim)).astype(np.float32)
    idx.set_pq(cb)
    idx.set_coarse(coarse)
    if shard is not None:
        shard["init"](idx)
    for kv in filter(None, os.environ.get("QADC_BENCH_IVF_OPTS", "").split(",")):  # tuning experiments only
        idx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
    qs = [rng.normal(size=(NQB, dim)).astype(np.float32) for _ in range(8)]

    def collect(slot, nq):
        """-> probed codes of the batch (whole database: every rank counts the same figure)."""
        if shard is None:
            return int(sizes[idx.search_collect(slot)["assign"]].sum())
        idx.dist_collect(slot)
        return int(sizes[idx.slot_assign(slot, nq, MA)].sum())

    def pipelined(batches, steps, depth=3, warm=True):
        nqb = batches[0].shape[0]
        for w in range(depth):                             # every slot used below sizes its buffers before the clock starts
            idx.search_submit(w, batches[w], MA, R)
        for w in range(depth):
            collect(w, nqb)
        if warm:                                           # ... and one untimed pass in the steady state of the loop below
            pipelined(batches, 2 * depth, depth, warm=False)   # (runtime-side lazy growth under a full pipeline was seen to
        idx.profile_reset()                                #  stall a first timed pass by tens of milliseconds)
        if shard is not None and shard.get("barrier"):
            shard["barrier"]()
        t0 = time.perf_counter()
        pend, nc = [], 0
        stamps = []
        t_sub = t_col = 0.0
        for s in range(steps):
            stamps.append(time.perf_counter())
            idx.search_submit(s % depth, batches[s % len(batches)], MA, R)
            t_sub += time.perf_counter() - stamps[-1]
            pend.append(s % depth)
            if len(pend) == depth:
                tc = time.perf_counter()
                nc += collect(pend.pop(0), nqb)
                t_col += time.perf_counter() - tc
        if os.environ.get("QADC_BENCH_IVF_STEPLOG"):
            print("ivf steps (ms):", [round((b - a) * 1e3, 2) for a, b in zip(stamps, stamps[1:])], file=sys.stderr)
            print("host ms per step: submit %.3f collect %.3f" % (t_sub * 1e3 / steps, t_col * 1e3 / steps), file=sys.stderr)
        while pend:
            nc += collect(pend.pop(0), nqb)
        if shard is not None and shard.get("barrier"):
            shard["barrier"]()
        return time.perf_counter() - t0, nc

    cpu_sample = None
    if shard is None and float(os.environ.get("QADC_BENCH_CPU_SECONDS", 15)) > 0:
        # (untimed) the first queries of the first batch once more, synchronously, with everything the CPU leg needs: assign[],
        # the device's int8 tables, the heaps
        NS = int(os.environ.get("QADC_BENCH_IVF_CPU_QUERIES", 32))
        r0 = idx.search(qs[0], MA, R)
        cpu_sample = dict(M=M, sizes=sizes, seed0=seed0, assign=r0["assign"][:NS].copy(), qtables=idx.slot_qtables(0, 0, NS, MA),
                          keys=r0["keys"][:NS].copy(), values=r0["values"][:NS].copy(), heap_sizes=r0["sizes"][:NS].copy())
    steps, depth = 48, int(os.environ.get("QADC_BENCH_IVF_DEPTH", 4))          # all four submission slots of the C-ABI in use
    dt, ncodes = pipelined(qs, steps, depth)
    p = idx.profile()
    qs2 = [rng.normal(size=(2 * NQB, dim)).astype(np.float32) for _ in range(8)]       # and at twice the batch size
    dt2, _ = pipelined(qs2, 24, depth)
    # a short pass with the library's HIP events on (they cost ~10 us of stream time each: not in the timed loops above):
    # the launches of the partition-major second phase against THEIR roofs
    idx.set_option("profile", 1)
    dtp, _ = pipelined(qs, 16, depth, warm=False)
    pp = idx.profile()
    # ... and the same launches ALONE on the GPU (one batch at a time: nothing of another batch runs beside them) — the kernel's
    # own duration, what `rocprofv3 --pmc` passes (which serialise kernels) see; the pipelined figures above include whatever
    # the neighbouring batches' fronts, orderings and replays take from the launch
    pa = None
    if pp["group_batches"] and shard is None:
        idx.profile_reset()
        pipelined(qs, 8, 1, warm=False)
        pa = idx.profile()
    idx.set_option("profile", 0)
    idx.close()
    roof = None
    if pp["group_batches"]:
        nb = pp["group_batches"]
        lds_cycles = (pp["group_pass_codes8"] * M * 4 + pp["group_pass_codes4"] * M * 2) / 64.0
        lds_rate = lds_cycles / (pp["group_scan_ms"] * 1e-3) / 1e9
        head_gbs = pp["group_head_codes"] * (M // 2) / (pp["group_head_ms"] * 1e-3) / 1e9
        # what the head LAUNCH has to read besides the head probes' codes (the front is part of the path: SURVEY.md 8 rows A5-A7): the
        # starts of ALL probed partitions for the float pre-scan (max(1, size x keep) codes each: keep x the probed codes, to within
        # the rounding of ~10^2-code start runs) and every probe's float table once
        front_bytes_per_query = KEEP * (ncodes / (steps * NQB)) * (M // 2) + MA * M * 16 * 4
        head_alg_incl = pp["group_head_codes"] * (M // 2) / nb + front_bytes_per_query * NQB
        head_gbs_incl = head_alg_incl / (pp["group_head_ms"] / nb * 1e-3) / 1e9
        roof = {"bound": "lds", "kernel": "scan_i8_mq_narrow_kernel<%d,2> (8- and 4-seat groups) over the (query, probe) pairs regrouped by partition" % M,
                "achieved": lds_rate, "peak": LDS_PEAK_GCYC, "unit": "G LDS-array cycles/s", "frac": lds_rate / LDS_PEAK_GCYC,
                "avg_launch_ms": pp["group_scan_ms"] / nb, "launches": nb,
                "lds_cycles_rule": "codes read by 8-seat passes x %d lookups x 4 cycles / 64 lanes + codes read by 4-seat passes x %d x 2 / 64 "
                                   "(a group of <= 4 pairs takes 8-byte rows); recounted from assign[] on the host" % (M, M),
                "seat_fill": pp["group_pairs"] / max(pp["group_seats"], 1), "pairs_per_batch": pp["group_pairs"] / nb,
                "codes_read_per_batch": (pp["group_pass_codes8"] + pp["group_pass_codes4"]) / nb,
                "head": {"bound": "hbm", "kernel": "scan_query_kernel<%d, HEAD> (front: pre-scan, select, quantizer of all %d tables; then the "
                                                   "first probes of every query, one workgroup per query)" % (M, MA),
                         "achieved": head_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": head_gbs / HBM_PEAK_GBS,
                         "avg_launch_ms": pp["group_head_ms"] / nb,
                         "algorithmic_bytes_per_launch": pp["group_head_codes"] * (M // 2) / nb,
                         "note": "bytes = M/2 x codes of the head probes only; the launch also pre-scans the starts of all probes and "
                                 "quantizes their tables, so frac understates the walk",
                         "incl_front_reads": {
                             "achieved": head_gbs_incl, "frac": head_gbs_incl / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": head_alg_incl,
                             "rule": "head probes' codes + per query: keep x probed codes x M/2 B (the starts of all %d probes, read by the float "
                                     "pre-scan: scan_4 over the starts, query_common.hpp:59-90) + %d float tables of %d B — the bytes the launch "
                                     "cannot avoid reading; its PMC traffic (profiles/r05_ivf*_pmc_summary.json) also holds the value scratch "
                                     "and the int8 tables it writes" % (MA, MA, M * 64)}},
                "launches_alone_on_the_gpu": None if not (pa and pa["group_batches"]) else {
                    "what": "the same three launches with ONE batch in flight (8 batches): each kernel by itself, as a PMC pass sees it",
                    "head_ms": pa["group_head_ms"] / pa["group_batches"], "grouped_scan_ms": pa["group_scan_ms"] / pa["group_batches"],
                    "order_ms": pa["group_order_ms"] / pa["group_batches"],
                    "head_frac_of_hbm": pa["group_head_codes"] * (M // 2) / (pa["group_head_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "head_frac_of_hbm_incl_front_reads": (pa["group_head_codes"] * (M // 2) / pa["group_batches"] + front_bytes_per_query * NQB)
                    / (pa["group_head_ms"] / pa["group_batches"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "grouped_scan_frac_of_lds": (pa["group_pass_codes8"] * M * 4 + pa["group_pass_codes4"] * M * 2) / 64.0
                    / (pa["group_scan_ms"] * 1e-3) / 1e9 / LDS_PEAK_GCYC},
                "order_cands_avg_launch_ms": pp["group_order_ms"] / nb,
                "timed_region": "%d profiled batches after the timed loops (HIP events on the library's stream around each launch); "
                                "wall %.3f ms per batch in that pass" % (nb, dtp * 1e3 / 16)}
    gbs = ncodes * (M // 2) / dt / 1e9
    out = {"workload": "IVF, %d x %dx4 codes (%d-d vectors) in K=%d partitions (multinomial sizes), nprobe=%d, R=%d, keep=%.0f%%, "
                       "%d-query batches, %d in flight, queries in -> heaps out (qadc_search)" % (N, M, dim, K, MA, R, KEEP * 100, NQB, depth),
           "codes_per_sec": ncodes / dt, "us_per_query": dt * 1e6 / (steps * NQB), "queries_per_sec": steps * NQB / dt,
           "ms_per_batch": dt * 1e3 / steps,
           "us_per_query_at_2048_query_batches": dt2 * 1e6 / (24 * 2 * NQB),
           "probed_codes_per_query": ncodes / (steps * NQB),
           "algorithmic_GBps": gbs,
           "algorithmic_GBps_rule": "M/2 B x probed codes / wall time of the pipelined batches (whole path, not one kernel; "
                                    "NOT an HBM figure: the partition-major second phase reads a partition once for up to 8 queries)",
           "batches_through_partition_major_second_phase": int(p["group_launches"]), "of_them_redone_on_the_level_path": int(p["group_fallbacks"]),
           "host_ms_per_batch": {"plan": p["host_plan_ms"] / steps, "stream_assembly": p["host_replay_ms"] / steps,
                                 "heap": p["host_heap_ms"] / steps}}
    if roof is not None:
        out["roofline"] = roof
    if cpu_sample is not None:
        out["_cpu_sample"] = cpu_sample
    if shard is not None:
        out.update({"rccl_ranks": shard["world"], "placement": shard["placement"], "codes_on_this_rank": local_codes,
                    "merge": shard.get("merge", "native: qadc_dist_collect"), "candidates_per_query_this_rank": p["candidates"] / (steps * NQB)})
    return out


def c2_leg(local_rank):
    """BASELINE configs[1]: flat 10M x 16x4 (80 MB) on one GPU, both scan modes — 32-query steps (8 queries per pass, three
    steps in flight) and ONE query per pass (the reference's mode), each in its own timed region.  The list is 80 MB: after
    the first pass it is served from the 256 MiB Infinity Cache / L2, and one query is a dependent chain of a head launch and
    two bound-level launches of a few microseconds of streaming each, so neither mode is HBM-bound here; `roofline_c2` reports
    the achieved algorithmic GB/s against the HBM peak anyway and says what bounds it."""
    import torch
    import pyqadc
    M, N, NQ = 16, int(float(os.environ.get("QADC_BENCH_C2_CODES", 1e7))), 32
    idx = pyqadc.Index(M, local_rank)
    idx.add_partition_synthetic(N, SEED + 2)
    idx.finalize(KEEP)
    for kv in filter(None, os.environ.get("QADC_BENCH_C2_OPTS", "").split(",")):   # tuning experiments only
        idx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
    rng = np.random.default_rng(4321)
    cb = rng.normal(size=(M, 16, 128 // M)).astype(np.float32)
    pool = [make_tables(rng, cb, NQ) for _ in range(4)]
    assign = np.zeros((NQ, 1), np.int32)

    def run(k):
        pend = []
        for s in range(k):
            idx.submit(s % 3, assign, pool[s % len(pool)].copy(), R)
            pend.append(s % 3)
            if len(pend) == 3:
                idx.collect(pend.pop(0))
        while pend:
            idx.collect(pend.pop(0))

    steps = 300
    run(12)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cs = M // 2
    # ---- ONE synchronous query per call, nothing else in flight: what nns_engine issues on this configuration (query_common.hpp:
    # 278-307) — call to return: the front sliced over workgroups in a launch of its own, then one walk launch of 32 workgroups ----
    sync_ts, _, _ = sync_query_us(idx, np.zeros((1, 1), np.int32), pool[0][:1].copy(), 150)
    idx.set_option("profile", 1)
    # ---- the metric's mode in the headline's FORM: 32-query steps, every query walks the 80 MB list by itself (one C call per
    # step, 32 independent runs per bound-level launch, no sibling launch, no multi-query pass), three steps in flight ----
    psteps = 100
    pprof, pdt = one_pass_batch_leg(idx, pool, psteps, warm=6)
    # the streaming ceiling of the same launches on this box (PROBE variant: lookups replaced by an XOR of the loaded words)
    idx.set_option("variant", 0x0d | 16)
    cprof, _ = one_pass_batch_leg(idx, pool, 30, warm=3)
    idx.set_option("variant", 0x0d)
    # ---- one query per CALL: sequential single-query batches, HIP events around the streaming launches ----
    set_mode(idx, MODE_BATCHED)
    idx.set_option("front_run_max", 0)
    nsingle, depth1 = 512, int(os.environ.get("QADC_BENCH_C2_DEPTH", 8))     # (a query is a chain of ~10 short launches: eight in flight)
    sprof, sdt = single_query_leg(idx, M, N, pool, nsingle, depth1)
    idx.close()
    nq_pass = NQ * psteps
    wall_gbs = float(N) * cs * nq_pass / pdt / 1e9
    kern_gbs = pprof["scan_codes"] * cs / (pprof["scan_ms"] * 1e-3) / 1e9 if pprof["scan_ms"] > 0 else 0.0
    ceil_gbs = cprof["scan_codes"] * cs / (cprof["scan_ms"] * 1e-3) / 1e9 if cprof["scan_ms"] > 0 else 0.0
    roof = {"bound": "infinity_cache", "achieved": kern_gbs, "peak": ceil_gbs, "unit": "GB/s", "frac": kern_gbs / ceil_gbs if ceil_gbs else None,
            "traffic": None,
            "peak_rule": "MEASURED: the same launches with every table lookup replaced by an XOR of the loaded words (PROBE variant), "
                         "%d launches, avg %.4f ms — the 80 MB list fits the 256 MiB Infinity Cache and every query re-reads it from there, "
                         "so HBM does not bound this configuration and the guide has no spec figure for the cache's streaming rate"
                         % (cprof["scan_launches"], cprof["scan_ms"] / max(cprof["scan_launches"], 1)),
            "achieved_rule": "%d B x codes of the event-timed bound-level launches / their HIP-event time (%d launches, avg %.4f ms)"
                             % (cs, pprof["scan_launches"], pprof["scan_ms"] / max(pprof["scan_launches"], 1)),
            "frac_of_hbm_peak_wall": wall_gbs / HBM_PEAK_GBS, "GBps_wall": wall_gbs,
            "ms_per_query_wall": pdt * 1e3 / nq_pass, "codes_per_sec_wall": float(N) * nq_pass / pdt,
            "kernel": "scan_i8_kernel<%d,2,cached,chunk> (%d queries per launch, one pass over the level's codes each)" % (M, NQ),
            "mode": "one query per pass over the list (simd_scan.hpp:125-187 called once per query, db_query_4.cpp:287-308), "
                    "%d-query steps through ONE C call each" % NQ}
    return {"workload": "flat DB, %d x %dx4 PQ codes (%d MB), R=%d, keep=%.0f%%: BASELINE configs[1]" % (N, M, N * cs // 1000000, R, KEEP * 100),
            "batched": {"queries_per_step": NQ, "steps": steps, "ms_per_step": dt * 1e3 / steps, "codes_per_sec": float(N) * NQ * steps / dt,
                        "mode": "32 queries per step share every pass over the codes (8 per pass), three steps in flight"},
            "one_query_per_pass": {"codes_per_sec": float(N) * nq_pass / pdt, "ms_per_query": pdt * 1e3 / nq_pass, "queries": nq_pass,
                                   "ms_per_step": pdt * 1e3 / psteps, "queries_per_step": NQ,
                                   "mode": "the headline's form: %d independent queries per launch, each walking the list by itself" % NQ},
            "one_query_per_call": {"codes_per_sec": float(N) * nsingle / sdt, "ms_per_query": sdt * 1e3 / nsingle, "queries": nsingle,
                                   "in_flight": depth1,
                                   "mode": "sequential single-query calls (a dependent chain of ~10 short launches each: launch latency bound)",
                                   "streaming_launch_avg_ms": sprof["scan_ms"] / max(sprof["scan_launches"], 1)},
            "one_query_synchronous": {"us_per_call": float(np.median(sync_ts)), "p10": float(sync_ts[len(sync_ts) // 10]),
                                      "p90": float(sync_ts[len(sync_ts) * 9 // 10]), "calls": int(len(sync_ts)),
                                      "mode": "ONE synchronous qadc_query_scan per call, call to return, nothing else in flight (the "
                                              "reference's per-query loop on this configuration); one thread of the reference's scan_avx_4 "
                                              "needs codes / cpu_baseline.value for the scan alone"}}, roof


def real_encode_recall_ivf(local_rank):
    """Recall@R of the IVF path on REAL encodings at BASELINE configs[2]'s proportions: K = 4096 cells,
    nprobe = 32 (0.8 % of the cells), clustered synthetic 128-d vectors; coarse centroids = sampled base vectors refined by two
    k-means rounds on a sample (qadc_kmeans_iterations_host), residuals PQ-encoded on the GPU (qadc_ivf_encode_host), partitions
    with labels = vector ids; queries through qadc_search; ground truth = exact float L2 nearest neighbour (torch, chunked).
    `reference_heaps_equal`: the reference's scan_avx_4 over the probed partitions (labels, one shared heap) with the device's
    int8 tables ends in the same heaps — this recall IS the reference path's recall on the same tables."""
    import torch
    import pyqadc
    M, dim, K, MA, nq = 16, 128, 4096, 32, 256
    n = int(float(os.environ.get("QADC_BENCH_REAL_IVF_CODES", 4e6)))
    dev = torch.device("cuda", local_rank)
    g = torch.Generator(device=dev).manual_seed(11)
    C = max(2000, n // 300)
    centres = torch.randn(C, dim, device=dev, generator=g) * 3
    base = centres[torch.randint(0, C, (n,), device=dev, generator=g)] + torch.randn(n, dim, device=dev, generator=g)
    queries = centres[torch.randint(0, C, (nq,), device=dev, generator=g)] + torch.randn(nq, dim, device=dev, generator=g)
    best_d = torch.full((nq,), float("inf"), device=dev)
    best_i = torch.zeros(nq, dtype=torch.long, device=dev)
    for lo in range(0, n, 1 << 20):
        d = torch.cdist(queries, base[lo:lo + (1 << 20)])
        dmin, imin = d.min(dim=1)
        upd = dmin < best_d
        best_d = torch.where(upd, dmin, best_d)
        best_i = torch.where(upd, imin + lo, best_i)
    gt = best_i.cpu().numpy()
    base_h = base.cpu().numpy()
    qh = queries.cpu().numpy()
    del base, queries, centres
    torch.cuda.synchronize()
    rng = np.random.default_rng(21)
    sample = base_h[rng.choice(n, min(n, 200000), replace=False)]
    coarse, _ = pyqadc.kmeans_iterations(sample, sample[rng.choice(sample.shape[0], K, replace=False)], 2, local_rank)
    coarse = np.where(np.isnan(coarse), sample[:K], coarse).astype(np.float32)      # (an empty cell keeps a sample point)
    ds = dim // M
    assign0, _ = pyqadc.ivf_encode(np.zeros((M, 16, ds), np.float32), sample[:50000], coarse=coarse, device=local_rank)
    resid = sample[:50000] - coarse[assign0]
    cb = np.stack([resid[rng.choice(resid.shape[0], 16, replace=False), m * ds:(m + 1) * ds] for m in range(M)]).astype(np.float32)
    assign = np.zeros(n, np.int32)
    codes = np.zeros((n, M // 2), np.uint8)
    for lo in range(0, n, 1 << 20):                             # db_add's chunks (db_add.cpp:52-82)
        a, c = pyqadc.ivf_encode(cb, base_h[lo:lo + (1 << 20)], coarse=coarse, device=local_rank)
        assign[lo:lo + len(a)], codes[lo:lo + len(a)] = a, c
    order = np.argsort(assign, kind="stable")                   # vector order inside a partition = id order (databases.hpp:291-297)
    bounds = np.searchsorted(assign[order], np.arange(K + 1))
    parts = [codes[order[bounds[k]:bounds[k + 1]]] for k in range(K)]
    labels = [order[bounds[k]:bounds[k + 1]].astype(np.uint32) for k in range(K)]
    idx = pyqadc.Index(M, local_rank)
    idx.add_partitions(parts, labels)
    idx.finalize(KEEP)
    idx.set_pq(cb)
    idx.set_coarse(coarse)
    res = idx.search(qh, MA, R)
    hits = sum(int(gt[q] in set(res["keys"][q][:res["sizes"][q]].tolist())) for q in range(nq))
    probed = float(np.mean([sum(parts[p].shape[0] for p in res["assign"][q]) for q in range(nq)]))
    out = {"value": hits / nq, "codes": n, "queries": nq, "K": K, "nprobe": MA, "probed_codes_per_query": probed,
           "queries_with_status_1": int((res["status"] != 0).sum()),
           "data": "synthetic 128-d vectors, %d clusters (3*N(0,1) centres + N(0,1)); coarse centroids: %d sampled vectors + 2 k-means "
                   "rounds on a 200 K sample; codebooks = sampled residual sub-vectors; residuals PQ %dx4 encoded on the GPU; labels = "
                   "vector ids; ground truth = exact float L2 NN" % (C, K, M)}
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    if po.have_ref() and float(os.environ.get("QADC_BENCH_CPU_SECONDS", 15)) > 0:
        NS = 64
        qt = idx.slot_qtables(0, 0, NS, MA)
        inter = {}
        same, hits_ref = 0, 0
        for q in range(NS):
            if res["status"][q]:
                same += 1                                       # (the reference would have exited; nothing to compare)
                continue
            ps = [int(p) for p in res["assign"][q]]
            for p in ps:
                if p not in inter:
                    inter[p] = po.ref_interleave(parts[p]) if parts[p].shape[0] else np.zeros(0, np.uint8)
            k, v = po.ref_scan_interleaved(M, [inter[p] for p in ps], [parts[p].shape[0] for p in ps], [labels[p] for p in ps], qt[q], R)
            sz = int(res["sizes"][q])
            same += int(np.array_equal(k, res["keys"][q, :sz]) and np.array_equal(v, res["values"][q, :sz]))
            hits_ref += int(gt[q] in set(k.tolist()))
        out["reference_heaps_equal"] = "%d/%d" % (same, NS)
        out["reference_recall_at_100_first_%d" % NS] = hits_ref / NS
    idx.close()
    return out


def sync_query_us(idx, a, tb, reps, warm=20):
    """Sorted call-to-return times (us) of `reps` synchronous qadc_query_scan calls of ONE query (assign a [1][ma], float tables
    tb), the C-ABI call as a C/C++ caller makes it: caller-owned output buffers, no per-call allocation on the Python side, a fresh
    copy of the float tables per call (the call may clamp them in place).  Also returns the last call's (keys, values)."""
    import pyqadc
    keys, vals = np.zeros((1, R), np.uint32), np.zeros((1, R), np.int8)
    sizes, status = np.zeros(1, np.int32), np.zeros(1, np.int32)
    qmin, qmax = np.zeros(1, np.float32), np.zeros(1, np.float32)
    P = pyqadc._p
    fixed = (P(keys, pyqadc.u32p), P(vals, pyqadc.i8p), P(sizes, pyqadc.i32p), P(status, pyqadc.i32p),
             P(qmin, pyqadc.f32p), P(qmax, pyqadc.f32p), None)
    pa = P(a, pyqadc.i32p)
    fn, h = pyqadc.lib().qadc_query_scan, idx._h
    copies = [tb.copy() for _ in range(reps + warm)]
    ptrs = [P(t, pyqadc.f32p) for t in copies]
    ts = []
    for i in range(reps + warm):
        t0 = time.perf_counter()
        rc = fn(h, 1, a.shape[1], pa, ptrs[i], R, *fixed)
        t1 = time.perf_counter()
        assert rc == 0 and status[0] == 0 and sizes[0] == R
        if i >= warm:
            ts.append(t1 - t0)
    return np.sort(np.array(ts)) * 1e6, keys, vals


def latency_leg(local_rank):
    """Synchronous single query (nq = 1) on a 10^5-code flat list: the reference's only published Quick-ADC point is
    86 us of scan time on ~93 750 probed codes, one CPU thread (README.md:327-330)."""
    import pyqadc
    M, n = 16, 100000
    idx = pyqadc.Index(M, local_rank)
    idx.add_partition_synthetic(n, 1)
    idx.finalize(KEEP)
    rng = np.random.default_rng(0)
    cb = rng.normal(size=(M, 16, 8)).astype(np.float32)
    tb = make_tables(rng, cb, 1)
    a = np.zeros((1, 1), np.int32)
    for _ in range(20):
        idx.query_scan(a, tb.copy(), R)
    # the C-ABI call as a C/C++ caller makes it (sync_query_us; pyqadc.Index.query_scan allocates six arrays and builds Python
    # tuples per call: ~7 us of interpreter time that is not the library's)
    ts, keys, vals = sync_query_us(idx, a, tb, 200)
    ts_py = []
    want = idx.query_scan(a, tb.copy(), R)                      # same answer as the allocating wrapper
    assert np.array_equal(want["keys"], keys) and np.array_equal(want["values"], vals)
    for _ in range(100):
        t = tb.copy()
        t0 = time.perf_counter()
        idx.query_scan(a, t, R)
        ts_py.append(time.perf_counter() - t0)
    idx.close()
    return {"value": float(np.median(ts)), "p10": float(ts[len(ts) // 10]), "p90": float(ts[len(ts) * 9 // 10]),
            "unit": "us", "codes": n, "through_allocating_python_wrapper": float(np.median(ts_py) * 1e6),
            "note": "synchronous qadc_query_scan (C-ABI, caller-owned buffers, called through ctypes), nq=1, R=100, "
            "keep=1%, float tables in -> heap out, host call to host return"}


# --------------------------------------------------------------------------------------------- launcher
def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves — as CHILD processes, from a parent
    that never touches the GPU (no exec of a GPU-initialised process) — and relay rank 0's JSON line."""
    if under_profiler():
        # a profiler's preloaded library has initialised the GPU in THIS process: spawning ranks from it is the forbidden
        # exec-after-GPU-init on this pool.  Profile one rank (`rocprofv3 ... -- python3 bench.py`) or launch the ranks first.
        raise SystemExit("bench.py: --gpus %d under a profiler: launch the ranks with torch.distributed.run and profile those" % args.gpus)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup)]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None)
    line = None
    for raw in proc.stdout:
        txt = raw.decode(errors="replace")
        if txt.startswith("{") and line is None:
            line = txt.strip()
        else:
            sys.stderr.write(txt)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)                                # (also from a launch that ended non-zero: rank 0's line says what was abandoned)
    if rc != 0 or line is None:
        raise SystemExit("bench.py: the %d-rank launch failed (exit code %d)" % (args.gpus, rc))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pmc-leg", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_leg:
        return pmc_leg_main()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args)

    line_printed = []                                          # (non-empty once rank 0 has printed the JSON line)
    import threading
    print_lock = threading.Lock()                              # the line is printed once: by main(), or by the watchdog of the N-rank legs
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (or omit the launcher and let "
                         "bench.py start the ranks itself)" % (args.gpus, world))
    M = int(os.environ.get("QADC_BENCH_M", 16))
    N = int(float(os.environ.get("QADC_BENCH_CODES", 1e9)))
    NQ = int(os.environ.get("QADC_BENCH_NQ", 32))
    cs = M // 2
    backend = os.environ.get("QADC_BENCH_BACKEND", "nccl")

    # ---- in-run HBM traffic of the one-query-per-pass mode: child processes under rocprofv3, BEFORE this process
    # touches the GPU (single rank only) ----
    pmc, pmc_src = None, "skipped"
    pmc32, pmc32_src = None, "skipped"
    if world == 1 and os.environ.get("QADC_BENCH_PMC", "1") != "0" and not os.environ.get("QADC_BENCH_FORCE_DIST"):
        pmc, pmc_src = pmc_traffic_in_run(M, N, NQ)
        if os.environ.get("QADC_BENCH_32X4", "1") != "0" and M == 16:
            pmc32, pmc32_src = pmc_traffic_in_run(32, N, 8)

    import torch
    import torch.distributed as dist
    import pyqadc
    from pyqadc import sharded
    # The interpreter's cyclic garbage collector stays off for the rest of the run: with torch loaded a full collection
    # walks a few hundred thousand objects (~50 ms — sixty 1024-query IVF batches, or forty steps of a 1/8 shard), and
    # when it fires depends on how many objects earlier legs allocated.  The loops below create no reference cycles.
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the Quick-ADC engine has no CPU path")
    # test hooks (not used by the driver): run the multi-rank path on a 1-GPU box over gloo
    if os.environ.get("QADC_BENCH_ONE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # The library's per-device stream set BEFORE any communicator exists (qadc_device_prepare): a communicator created first takes
    # one of the runtime's four highest-priority queues and the set's streams then land on shared queues / pipes — one of 8 ranks'
    # IVF batch measured 0.72 instead of 0.44 ms (C3) and 1.08 instead of 0.78 (C5): profiles/r05_queue_map_rccl.txt.  (torch's
    # own HIP runtime has to be up first — tests/conftest.py — hence not earlier than this.)
    torch.zeros(1, device=dev)
    pyqadc.device_prepare(local_rank)
    stream_layout = None
    cdev = dev if backend == "nccl" else torch.device("cpu")   # device of the collective buffers
    # QADC_BENCH_FORCE_DIST=1 takes the multi-rank code path (collectives included) even with one rank
    use_dist = world > 1 or bool(os.environ.get("QADC_BENCH_FORCE_DIST"))
    if use_dist:
        if world == 1:                                         # QADC_BENCH_FORCE_DIST without a launcher: rendezvous with ourselves
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: process group has %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus))

    # the deployment check of DESIGN.md section 5, with every communicator of this process alive: do the library's streams sit where
    # the measured figures assume (qadc_stream_layout: ten probes, ~3 ms)?
    stream_layout = pyqadc.stream_layout(local_rank)
    # ---- database: this rank's contiguous shard of the synthetic list + replica of the starts ----
    # measurement hook (tools/dist_sizes3.sh): ONE process stands in for rank 0 of LOOP ranks — it holds 1/LOOP of the list,
    # pre-scans 1/LOOP of the starts, and the loopback transport hands it LOOP copies of its own block to merge
    LOOP = int(os.environ.get("QADC_BENCH_LOOPBACK_WORLD", 0)) if world == 1 and use_dist else 0
    mworld = LOOP or world                                     # ranks the merge sees
    first, local_n = sharded.shard_ranges(N, mworld)[rank]
    starts = max(1, int(np.float32(N) * np.float32(KEEP)))
    idx = pyqadc.Index(M, local_rank)
    idx.add_partition_synthetic_shard(N, first, local_n, SEED, starts)
    idx.finalize(KEEP)
    idx.set_option("profile", 1)
    set_mode(idx, MODE_ONE_QUERY_PER_PASS)                     # the headline's mode (the metric's: SURVEY.md 8d)
    for kv in filter(None, os.environ.get("QADC_BENCH_OPTS", "").split(",")):     # tuning experiments only
        idx.set_option(kv.split("=")[0], float(kv.split("=")[1]))

    # ---- native RCCL merge inside the library (default for real multi-rank runs): rank 0's unique id travels over the
    # process group that is up already; any failure on any rank sends ALL ranks to the torch.distributed path ----
    native_dist = False

    def native_init(ix):
        """qadc_dist_init on every rank with rank 0's id (the ranks agreed beforehand that RCCL loads everywhere)."""
        uid = torch.from_numpy(pyqadc.dist_unique_id() if rank == 0 else np.zeros(128, np.uint8)).to(dev)
        dist.broadcast(uid, 0)
        ix.dist_init(rank, world, uid.cpu().numpy())

    if use_dist and backend == "nccl" and os.environ.get("QADC_BENCH_NATIVE_DIST", "1") != "0":
        # agree on availability FIRST: a rank that cannot load librccl must not leave the others inside ncclCommInitRank
        ok = 1
        try:
            pyqadc.dist_unique_id()                            # loads RCCL in this process (the id itself is only used on rank 0)
        except Exception as e:  # noqa: BLE001
            sys.stderr.write("rank %d: native RCCL merge unavailable (%r), using the torch.distributed path\n" % (rank, e))
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        native_dist = bool(flag.item())
        if native_dist and LOOP:
            idx.dist_init_loopback(0, LOOP)
        elif native_dist:
            native_init(idx)
        for kv in filter(None, os.environ.get("QADC_BENCH_DIST_OPTS", "").split(",")) if native_dist else ():   # tuning experiments only
            idx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
    flat_tr = None
    run_nonce = os.environ.get("MASTER_PORT", "0")
    if use_dist and backend != "nccl" and os.environ.get("QADC_BENCH_NATIVE_DIST", "1") != "0":
        # test hook (ranks sharing one GPU, no RCCL between them): the SAME native merge over the library's shared-memory
        # transport, so that a 1-GPU box runs the multi-rank loop below exactly as an 8-GPU node does, collectives included
        # (segment names carry a per-run nonce agreed over the process group: a rerun on the same port after a crash must not
        # meet the crashed run's segment)
        nonce_t = torch.tensor([int.from_bytes(os.urandom(4), "little") & 0x7fffffff if rank == 0 else 0], dtype=torch.int64)
        dist.broadcast(nonce_t, 0)
        run_nonce = "%s_%08x" % (os.environ.get("MASTER_PORT", "0"), int(nonce_t.item()))
        flat_tr = pyqadc.ShmTransport("/qadc_bench_%s_flat" % run_nonce, rank, world)
        idx.dist_init_transport(flat_tr)
        native_dist = True

    rng = np.random.default_rng(1234)
    codebooks = rng.normal(size=(M, 16, 128 // M)).astype(np.float32)
    pool = [make_tables(rng, codebooks, NQ) for _ in range(4)]   # 4 distinct query batches, reused cyclically
    assign = np.zeros((NQ, 1), np.int32)

    def run_steps(k):
        """k pipelined steps, three batches in flight: while batch s scans, batch s-1 is collected and replayed on the
        host and the front of batch s+1 (upload, pre-scan, quantizer) already runs on its own stream."""
        if use_dist:
            return run_steps_dist(k)
        last = None
        pending = []
        for s in range(k):
            idx.submit(s % 3, assign, pool[s % len(pool)].copy(), R)
            pending.append(s % 3)
            if len(pending) == 3:
                last = idx.collect(pending.pop(0))
        while pending:
            last = idx.collect(pending.pop(0))
        return last

    def run_steps_dist(k):
        """Multi-rank steps.  Every rank pre-scans 1/world of the starts (the rest of the path is sharded by codes, the
        pre-scan by starts); ONE all-gather per step carries the finished batch's candidate streams and the pre-scan
        values of the batch three steps ahead.  Per iteration i: enqueue the sliced pre-scan of batch i+4, collect batch i,
        gather [streams of i | pre-scan values of i+3], replay, submit batch i+3 — batches i+1 and i+2 keep the GPU
        busy meanwhile (host jitter of a whole step is absorbed), and a pre-scan has a whole extra batch of lead (its
        kernels only find room at the boundaries of the long scan launches)."""
        last = None
        if k <= 0:
            return last
        tbs = {}
        LEAD = int(os.environ.get("QADC_BENCH_LEAD", 3))       # batches in flight besides the one being collected (<= 7: 8 slots)
        NS, NT = LEAD + 1, LEAD + 3                            # submission slots / table copies in rotation

        def prescan(b):                                        # batch b's sliced pre-scan -> pre-slot b % 2
            tbs[b % NT] = pool[b % len(pool)].copy()
            idx.prescan_submit(b % 2, assign, tbs[b % NT], R, rank, mworld)

        def merge(slot_i, pv):
            """Finished batch -> (keys, vals, sizes[, gathered pre-scan values]).  Native: qadc_dist_collect — ONE
            ncclAllGather straight from the device-resident streams + device-side replay, no host staging, no second
            collective.  Fallback (QADC_BENCH_NATIVE_DIST=0, gloo test hook): pyqadc/sharded.py over torch.distributed."""
            if native_dist:
                out = idx.dist_collect(slot_i, extra=None if pv is None else pv.reshape(-1))
                res3 = (out["keys"], out["values"], out["sizes"])
                if pv is None:
                    return res3
                g = out["extra"].reshape(mworld, NQ, -1)
                return res3 + (np.ascontiguousarray(g.transpose(1, 0, 2)).reshape(NQ, -1),)
            res = idx.collect_candidates(slot_i)
            return sharded.merge_batch(res, NQ, R, res["status"], cdev, extra=pv)

        nb = min(LEAD, k)                                      # the first batches: one stand-alone gather for all
        pvs = []
        for b in range(min(nb, 2)):                            # (both pre-scan slots busy from the start)
            prescan(b)
        for b in range(nb):
            pvs.append(idx.prescan_collect(b % 2))
            if b + 2 < nb:
                prescan(b + 2)
        g = sharded.gather_prescan(np.concatenate(pvs), cdev)
        if LOOP:
            g = np.tile(g, (1, LOOP))
        for b in range(nb):
            idx.submit(b % NS, assign, tbs[b % NT], R, prescan=g[b * NQ:(b + 1) * NQ])
        if k > LEAD:
            prescan(LEAD)
        stamps = [] if os.environ.get("QADC_BENCH_STEP_LOG") else None
        for i in range(k):                                     # batches i .. i+LEAD-1 are in flight; i is collected now
            if stamps is not None:
                stamps.append(time.perf_counter())
            ta = time.perf_counter()
            if i + LEAD + 1 < k:
                prescan(i + LEAD + 1)                          # pre-slot of batch i+LEAD-1, collected an iteration ago
            tb_ = time.perf_counter()
            pv = idx.prescan_collect((i + LEAD) % 2) if i + LEAD < k else None   # batch i+LEAD's, enqueued an iteration ago
            tc = time.perf_counter()
            out = merge(i % NS, pv)
            td = time.perf_counter()
            last = out[:3]
            if i + LEAD < k:
                idx.submit((i + LEAD) % NS, assign, tbs[(i + LEAD) % NT], R, prescan=out[3])
            te = time.perf_counter()
            if stamps is not None:
                acc = run_steps_dist.acc = getattr(run_steps_dist, "acc", [0.0] * 5)
                for j, v in enumerate((tb_ - ta, tc - tb_, td - tc, te - td)):
                    acc[j] += v
                acc[4] += 1
            if stamps is not None and te - ta > 5e-3 and rank == 0:
                print("SLOW iter %d: prescan %.2f collect_pv %.2f merge %.2f submit %.2f ms" % (i, (tb_-ta)*1e3, (tc-tb_)*1e3, (td-tc)*1e3, (te-td)*1e3), file=sys.stderr)
        if stamps and rank == 0:                                # tuning aid: per-iteration host times of the loop
            d = np.diff(np.array(stamps)) * 1e3
            acc = run_steps_dist.acc
            with open(os.environ["QADC_BENCH_STEP_LOG"], "a") as f:
                f.write("host ms per iteration: prescan_submit %.3f prescan_collect %.3f merge %.3f submit %.3f (profile: host_heap %.3f plan %.3f)\n" % (
                    acc[0] / acc[4] * 1e3, acc[1] / acc[4] * 1e3, acc[2] / acc[4] * 1e3, acc[3] / acc[4] * 1e3,
                    idx.profile()["host_heap_ms"] / max(acc[4], 1), idx.profile()["host_plan_ms"] / max(acc[4], 1)))
                run_steps_dist.acc = [0.0] * 5
                f.write("steps %d: median %.3f ms, max %.3f at %d, >2x median: %s\n" % (
                    k, np.median(d), d.max(), int(d.argmax()), [(int(j), round(float(d[j]), 2)) for j in np.nonzero(d > 2 * np.median(d))[0]][:40]))
        return last

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    idx.profile_reset()
    sync()
    t0 = time.perf_counter()
    last = run_steps(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = idx.profile()

    # ---- Recall@100 against the exact float-ADC nearest code, last batch, untimed ----
    tb = pool[(args.steps - 1) % len(pool)]
    keys = last[0] if use_dist else last["keys"]
    hits = 0
    for q in range(NQ):
        key, _, dist_q = idx.float_top1(0, tb[q, 0])
        if use_dist:
            cand = torch.tensor([dist_q, float(key)], dtype=torch.float64, device=cdev)
            allc = torch.empty(2 * world, dtype=torch.float64, device=cdev)
            dist.all_gather_into_tensor(allc, cand)
            allc = allc.cpu().numpy().reshape(world, 2)
            key = int(allc[np.lexsort((allc[:, 1], allc[:, 0]))[0], 1])   # min distance, lowest key on ties
        hits += int(key in set(keys[q].tolist()))
    recall = hits / NQ

    # ---- `roofline`: the headline's OWN timed region (one query per pass), this rank's shard; rank 0's goes in the line ----
    alg = prof["scan_codes"] * cs / max(prof["scan_launches"], 1)
    if pmc is not None:
        traffic, tsrc = pmc["traffic_over_algorithmic"] * alg, pmc_src + " (ratio of the same kind of launches x this region's bytes per launch)"
    elif world == 1 and not use_dist:
        traffic, tsrc2 = pmc_traffic_from_profiles(M, N, alg)
        tsrc = "%s; %s" % (pmc_src, tsrc2)
    else:
        traffic, tsrc = None, "no PMC child pass in a multi-rank run (the N=1 line carries it)"
    headline_roof = roofline_single(
        M, local_n, prof, elapsed, NQ * args.steps, traffic, tsrc,
        region="the headline's timed region: %d steps of %d queries, every query walks %s by itself (no sibling launch, no "
               "multi-query pass), three steps in flight; HIP events around every run of consecutive streaming launches on the "
               "library's stream" % (args.steps, NQ, "the whole list" if mworld == 1 else "this rank's 1/%d shard of the list" % mworld))
    headline_roof["kernel"] = "scan_i8_kernel<%d,2,nt,chunk> (%d queries per launch, one pass over the level's codes each)" % (M, NQ)
    if mworld > 1:
        headline_roof["shard"] = {"rank": rank, "of": mworld, "codes": local_n}
    if pmc is not None:
        headline_roof["pmc"] = pmc

    # ---- the measured streaming-read ceiling of this box ("peak ... and also a measured streaming-read ceiling"):
    # the SAME steps through the kernel's PROBE variant (variant bit 4: the LDS lookups replaced by an XOR of the loaded words —
    # the loads, the tiling and the launches are the headline's; the heaps are meaningless), untimed leg, own HIP events ----
    if world == 1 and not use_dist and os.environ.get("QADC_BENCH_CEILING", "1") != "0":
        idx.set_option("variant", 0x0d | 16)
        run_steps(1)
        idx.profile_reset()
        sync()
        run_steps(max(3, args.steps // 2))
        sync()
        cprof = idx.profile()
        idx.set_option("variant", 0x0d)
        if cprof["scan_ms"] > 0:
            ceil_gbs = cprof["scan_codes"] * cs / (cprof["scan_ms"] * 1e-3) / 1e9
            headline_roof["measured_ceiling_GBps"] = ceil_gbs
            headline_roof["frac_of_measured"] = headline_roof["achieved"] / ceil_gbs
            headline_roof["measured_ceiling_how"] = ("scan_i8_kernel<%d,2,nt,chunk,PROBE>: the headline's launches (same 7 bound levels x %d "
                                                     "queries, same grid, same 16-B non-temporal loads) with every table lookup replaced by an "
                                                     "XOR of the loaded words; %d launches, avg %.3f ms; HIP events on the library's stream"
                                                     % (M, NQ, cprof["scan_launches"], cprof["scan_ms"] / max(cprof["scan_launches"], 1)))

    # ---- the batched mode, reported separately: the same steps, queries as L2-sharing siblings, 8 per pass ----
    batched_elapsed, bprof = None, None
    if os.environ.get("QADC_BENCH_BATCHED", "1") != "0":
        set_mode(idx, MODE_BATCHED)
        run_steps(args.warmup)
        idx.profile_reset()
        sync()
        t0 = time.perf_counter()
        run_steps(args.steps)
        sync()
        batched_elapsed = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([batched_elapsed], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            batched_elapsed = float(t.item())
        bprof = idx.profile()
        set_mode(idx, MODE_ONE_QUERY_PER_PASS)

    # ---- one query per CALL (sequential single-query batches): the strictest reading of the reference's mode, single rank only ----
    single = None
    if world == 1 and not use_dist:
        nsingle = int(os.environ.get("QADC_BENCH_SINGLE_QUERIES", 64))
        if nsingle > 0:
            sprof, sdt = single_query_leg(idx, M, N, pool, nsingle)
            single = roofline_single(M, N, sprof, sdt, nsingle, None, "not collected for this leg (the headline's PMC passes cover the same kernel)")

    if rank == 0:
        total_codes = float(N) * NQ * args.steps
        out = {
            "metric": "pq_codes_scanned_per_sec", "value": total_codes / elapsed, "unit": "codes/s",
            "n_gpus": world, "rccl_ranks": dist.get_world_size() if use_dist else 1,
            "multi_gpu_merge": (("native: qadc_dist_collect (one ncclAllGather of device-resident push streams + device replay)"
                                 if backend == "nccl" else "native: qadc_dist_collect over the shared-memory transport (test hook)")
                                if native_dist else "pyqadc/sharded.py over torch.distributed (%s)" % backend) if use_dist else None,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "int8", "data": "synthetic",
            "config": {"workload": "flat DB, %d x %dx4 PQ codes (%d B/code), R=%d, keep=%.2f%%, sharded over %d GPU(s); a step = "
                                   "%d queries, ONE QUERY PER PASS: every query scans the whole list by itself (the reference's "
                                   "mode, HBM-bound: `roofline` is this same timed region).  The multi-query-per-pass mode of the "
                                   "same steps is reported separately as `value_batched` / `roofline_batched`"
                                   % (N, M, cs, R, KEEP * 100, world, NQ),
                       "codes": N, "M": M, "R": R, "keep": KEEP, "queries_per_step": NQ, "mode": "one query per pass",
                       "parallelism": "shard%d" % world},
            "recall_at_100": recall,
            "recall_note": "uniform random codes: the exact float-ADC nearest code is in the returned 100 for every query (near-vacuous); "
                           "the credible figures are `recall_at_100_real_encode(_ivf)` with `reference_heaps_equal`",
            "roofline": headline_roof,
            "phases": {"prescan_quantize_ms_per_step": prof["start_ms"] / args.steps,
                       "scan_kernel_ms_per_step": prof["scan_ms"] / args.steps,
                       "host_sort_replay_ms_per_step": prof["host_replay_ms"] / args.steps,
                       "candidates_per_query": prof["candidates"] / (NQ * args.steps), "regrows": prof["regrows"]},
            "stream_layout": stream_layout,
            "multi_gpu_note": "no 8-GPU node was available to the builder: N > 1 has only run as N processes on ONE GPU over the "
                              "shared-memory transport (tests) and with one rank over RCCL; no scaling curve was measured",
        }
        if single is not None:
            out["roofline_one_query_per_call"] = single
            out["value_one_query_per_call"] = single["codes_per_sec_wall"]
        if bprof is not None:
            scan_ms = bprof["scan_ms"]
            launches = max(bprof["scan_launches"], 1)
            mq = bprof["mq_launches"] > 0
            # LDS-array cycles the launches need (MI355X_MICROARCH.md, LDS): multi-query kernel = one ds_read_b128 (4 cycles
            # per 64 lanes) per code nibble and pass; single-query kernel = one ds_read_u8 (2 cycles) per code byte and query
            lds_cycles = bprof["pass_codes"] * (M * 4 if mq else (M // 2) * 2) / 64.0
            lds_rate = lds_cycles / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
            out["value_batched"] = total_codes / batched_elapsed
            out["ms_per_step_batched"] = batched_elapsed * 1e3 / args.steps
            out["roofline_batched"] = {
                "bound": "lds", "achieved": lds_rate, "peak": LDS_PEAK_GCYC, "unit": "G LDS-array cycles/s",
                "frac": lds_rate / LDS_PEAK_GCYC,
                "kernel": ("scan_i8_mq_kernel<%d,2> (8 queries per pass, %d passes per launch as L2-sharing siblings)"
                           % (M, (NQ + 7) // 8)) if mq else "scan_i8_kernel<%d,2> (sibling-major launch)" % M,
                "mode": "SURVEY.md 8f N2, its own timed region after the headline's: %d queries per step share every pass over the codes" % NQ,
                "launches": bprof["scan_launches"], "avg_launch_ms": scan_ms / launches,
                "lds_cycles_rule": "code reads of the launch x %d lookups x %d LDS cycles / 64 lanes "
                                   "(MI355X_MICROARCH.md LDS table); peak = 256 CUs x 2.4 GHz nominal — the PMC passes in "
                                   "profiles/ measure an effective clock of ~1.9-2.1 GHz under this load, i.e. the pipe is "
                                   "busier than frac says" % ((M, 4) if mq else (M // 2, 2)),
                "code_reads_per_launch": bprof["pass_codes"] / launches,
                "pair_rate": {"value": bprof["scan_codes"] / (scan_ms * 1e-3) if scan_ms > 0 else 0.0,
                              "unit": "(code, query) pairs/s inside the timed launches"},
                "hbm_note": "the queries of a launch share the codes through L2, so the HBM interface moves about 1/%d of "
                            "%d B x (code, query) — see profiles/r02_batched_hbm_traffic.json; this mode is NOT priced "
                            "against the HBM roofline (SURVEY.md 8d)" % (NQ, cs)}
    cpu_s = float(os.environ.get("QADC_BENCH_CPU_SECONDS", 15))
    if rank == 0 and world == 1 and not use_dist and cpu_s > 0:
        # the int8 tables of one bench batch, for the CPU legs (same tables on both sides)
        qt_cpu = idx.query_scan(assign, pool[0].copy(), R, want_qtables=True)["qtables"][:, 0]
    idx.close()
    if flat_tr is not None:
        flat_tr.close()
    # ---- BASELINE configs[2] / configs[4] shapes on N ranks: every rank holds its part of the partitions, the same query
    # batches go to every rank, qadc_dist_collect merges the push streams ----
    if use_dist and int(float(os.environ.get("QADC_BENCH_IVF_CODES", 1e8))) > 0:
        shm_tr = []

        def shard_init(ix):
            if native_dist and backend == "nccl":
                return native_init(ix)
            # no RCCL between these ranks (gloo test hook: the ranks share one GPU): the library's shared-memory transport
            tr = pyqadc.ShmTransport("/qadc_bench_%s_%d" % (run_nonce, len(shm_tr)), rank, world)
            shm_tr.append(tr)
            ix.dist_init_transport(tr)

        shard = dict(rank=rank, world=world, placement=os.environ.get("QADC_BENCH_IVF_PLACEMENT", "range"), init=shard_init,
                     barrier=sync, merge="native: qadc_dist_collect over " + ("RCCL" if native_dist and backend == "nccl" else "the shared-memory transport"))
        # The headline above is measured; these extra legs must not be able to take it down with them.  A watchdog on
        # EVERY rank: if the legs are not through in time (a collective that never completes), rank 0 prints the line without
        # them and every rank leaves — a hung rank cannot be interrupted, only abandoned.
        def give_up():
            # a hung collective cannot be interrupted, only abandoned — and the abandonment must be VISIBLE to the launcher: rank 0
            # prints the line (headline + the error) if the main thread has not, then every rank exits non-zero
            with print_lock:
                if rank == 0 and not line_printed:
                    out["ivf"] = {"error": "the multi-rank IVF legs did not finish within %s s; abandoned, exit code 3" % limit}
                    print(json.dumps(out), flush=True)
                    line_printed.append(1)
                if rank != 0:
                    time.sleep(1.0)                            # (rank 0's line first: a launcher tears the other ranks down at the first exit)
                os._exit(3)

        limit = float(os.environ.get("QADC_BENCH_IVF_TIMEOUT", 420))
        dog = threading.Timer(limit, give_up)
        dog.daemon = True
        dog.start()
        ivf_n, ivf_c5_n, ivf_failed = None, None, False
        try:
            ivf_n = ivf_leg(local_rank, shard=dict(shard))
            if os.environ.get("QADC_BENCH_IVF_C5", "1") != "0" and N >= 1e9:
                ivf_c5_n = ivf_leg(local_rank, M=32, K=16384, MA=64, dim=96, N=int(1e9), seed0=7000, shard=dict(shard))
        except Exception as e:  # noqa: BLE001 — reported in the line, the headline stands
            ivf_n = ivf_n or {"error": repr(e)}
            sys.stderr.write("rank %d: multi-rank IVF leg failed: %r\n" % (rank, e))
            ivf_failed = True                                  # (the other ranks may be stuck in a collective: the watchdog stays armed)
        if not ivf_failed:
            dog.cancel()
        for tr in shm_tr:
            tr.close()
        if rank == 0:
            if isinstance(ivf_n, dict) and "error" not in ivf_n:
                ivf_n["scaling_note"] = ("BASELINE configs[2] is a ONE-GPU configuration (100M codes, 0.8 ms per 1024-query batch on one "
                                         "GPU): a rank's batch is fixed costs (front, launches, merge), it is not expected to scale with "
                                         "the ranks; configs[4] (`ivf_c5`) is the multi-GPU IVF configuration")
            out["ivf"] = ivf_n
            if ivf_c5_n is not None:
                out["ivf_c5"] = ivf_c5_n
    if rank == 0 and world == 1 and not use_dist:
        torch.cuda.synchronize()
        qt32_cpu = None
        if os.environ.get("QADC_BENCH_32X4", "1") != "0" and M == 16:
            i32 = pyqadc.Index(32, local_rank)
            i32.add_partition_synthetic_shard(N, 0, N, SEED, starts)
            i32.finalize(KEEP)
            i32.set_option("profile", 1)
            cb32 = rng.normal(size=(32, 16, 4)).astype(np.float32)
            NQ32, steps32 = 16, 3
            pool32 = [make_tables(rng, cb32, NQ32)]
            p32, dt32 = one_pass_batch_leg(i32, pool32, steps32, warm=1)         # the headline's mode on the 32x4 list
            alg32 = p32["scan_codes"] * 16 / max(p32["scan_launches"], 1)
            out["roofline_32x4"] = roofline_single(32, N, p32, dt32, NQ32 * steps32, None if pmc32 is None else pmc32["traffic_over_algorithmic"] * alg32,
                                                   pmc32_src + ("" if pmc32 is None else " (ratio of the same kind of launches x this region's bytes per launch)"),
                                                   region="%d steps of %d queries, every query walks the whole list by itself (the headline's mode), three steps in "
                                                          "flight; HIP events around every run of consecutive streaming launches" % (steps32, NQ32))
            if pmc32 is not None:
                out["roofline_32x4"]["pmc"] = pmc32
            if cpu_s > 0:
                qt32_cpu = i32.query_scan(np.zeros((8, 1), np.int32), pool32[0][:8].copy(), R, want_qtables=True)["qtables"][:, 0]
            i32.close()
        cpu_samples = {}
        if int(float(os.environ.get("QADC_BENCH_IVF_CODES", 1e8))) > 0:
            out["ivf"] = ivf_leg(local_rank)
            cpu_samples["cpu_baseline_ivf"] = out["ivf"].pop("_cpu_sample", None)
            if "roofline" in out["ivf"]:
                out["roofline_ivf"] = out["ivf"].pop("roofline")
            if os.environ.get("QADC_BENCH_IVF_C5", "1") != "0" and N >= 1e9:
                # BASELINE configs[4] on ONE GPU: 1B x 32x4 codes (16 GB), 96-d vectors, nprobe 64
                out["ivf_c5_one_gpu"] = ivf_leg(local_rank, M=32, K=16384, MA=64, dim=96, N=int(1e9), seed0=7000)
                cpu_samples["cpu_baseline_ivf_c5"] = out["ivf_c5_one_gpu"].pop("_cpu_sample", None)
                if "roofline" in out["ivf_c5_one_gpu"]:
                    out["roofline_ivf_c5"] = out["ivf_c5_one_gpu"].pop("roofline")
        if os.environ.get("QADC_BENCH_C2", "1") != "0":
            out["c2"], out["roofline_c2"] = c2_leg(local_rank)         # BASELINE configs[1]: flat 10M x 16x4, both modes
        if os.environ.get("QADC_BENCH_LATENCY", "1") != "0":
            out["latency_us_single_query"] = latency_leg(local_rank)
        n_real = int(float(os.environ.get("QADC_BENCH_REAL_CODES", 1e7)))
        if n_real > 0:
            out["recall_at_100_real_encode"] = real_encode_recall(M, n_real, 64, local_rank)
            if int(float(os.environ.get("QADC_BENCH_REAL_IVF_CODES", 4e6))) > 0:
                out["recall_at_100_real_encode_ivf"] = real_encode_recall_ivf(local_rank)
        if cpu_s > 0:
            out["cpu_baseline"] = cpu_baseline(M, N, qt_cpu, cpu_s)
            out.update(cpu_extra_legs(M, N, qt_cpu, min(cpu_s, 6.0)))
            # the reference's CPU path beside the other legs (1 thread each, bounded): 32x4 flat, and the IVF legs' own queries
            if qt32_cpu is not None:
                out["cpu_baseline_32x4"] = cpu_baseline(32, N, qt32_cpu, min(cpu_s, 4.0))
            for key, smp in cpu_samples.items():
                if smp is not None:
                    out[key] = cpu_ivf_baseline(smp, min(cpu_s, 3.0))
    if rank == 0:
        with print_lock:
            if not line_printed:
                print(json.dumps(out), flush=True)
                line_printed.append(1)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
