"""Randomised parity sweep: random database shapes, batch shapes and tuning options (every kernel and fallback gets
hit by some seed), heaps compared bit for bit with the oracle's restatement of scanner_4::query_scan."""
import numpy as np
import pytest

from helpers import float_tables, heaps_equal, rand_codes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pyqadc():
    import pyqadc
    return pyqadc


import os


@pytest.mark.parametrize("seed", range(int(os.environ.get("QADC_FUZZ_SEEDS", 96))))
def test_random_configuration_matches_oracle(pyqadc, po, seed):
    rng = np.random.default_rng(1000 + seed)
    M = int(rng.choice([16, 32]))
    nparts = int(rng.integers(1, 5))
    sizes = [int(rng.choice([0, 1, 15, 16, 17, 100, 1000, 5000, 40000, 150001])) for _ in range(nparts)]
    if sum(sizes) == 0:
        sizes[0] = 33
    labelled = bool(rng.integers(0, 2))
    parts = [rand_codes(rng, n, M) for n in sizes]
    labels = [rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32) for n in sizes] if labelled else None
    R = int(rng.choice([1, 2, 10, 100, 257]))
    keep = float(rng.choice([0.001, 0.01, 0.05, 0.3]))
    same_assign = bool(rng.integers(0, 2))                 # identical probes for all queries: shared / multi-query launches
    nq = int(rng.integers(1, 21))
    ma = int(rng.integers(1, nparts + 1))
    if same_assign:
        assign = np.tile(rng.permutation(nparts)[:ma], (nq, 1)).astype(np.int32)
    else:
        assign = np.stack([rng.permutation(nparts)[:ma] for _ in range(nq)]).astype(np.int32)
    idx = pyqadc.Index(M)
    idx.add_partitions(parts, labels=labels)
    idx.finalize(keep)
    # every option of qadc_set_option that changes HOW a batch is computed (never what): tests/test_capi_host.py checks that this
    # list and the library's agree ("profile" and the qadc_dist_* ones apart: test_gpu_dist_multiproc.py draws those)
    opts = dict(small_run=int(rng.choice([1024, 8192, 131072])), mq=int(rng.integers(0, 2)),
                share_variant=int(rng.choice([0, 0x40, 0x48])), variant=int(rng.choice([0x00, 0x04, 0x08, 0x0c])),
                wgs_per_item=int(rng.choice([0, 0, 3])), cand_capacity=int(rng.choice([64, 1024, 16384])),
                prescan_sample=int(rng.choice([64, 4096, 65536])), level_base=int(rng.choice([64, 512, 4096])),
                level_growth=int(rng.choice([2, 4, 8])),
                front_run_max=int(rng.choice([0, 4096, 8 << 20])), device_replay_nq=int(rng.choice([0, 1, 1])), device_replay_alone_nq=int(rng.choice([0, 0, 512])),
                # one-workgroup-per-query path (`wgq` and `head_level` come from the scan_path parametrisation of tests/conftest.py):
                # workgroups per query, tiny stream / candidate capacities
                wgq_split=int(rng.choice([1, 3, 8])), wgq_split_codes=int(rng.choice([1024, 8192])),
                wgq_capacity=int(rng.choice([64, 4096])), wgq_cand_cap=int(rng.choice([64, 4096, 4096])),
                # partition-major second phase of device-replayed batches (with its overflow fallback), head length and form
                wgq_group=int(rng.choice([0, 2, 2])), wgq_group_head=int(rng.choice([1, 2, 4])), head_wg=int(rng.choice([0, 512])),
                # the float half: grouping of the pre-scan's adds and the quantizer as the reference binary has them, or as its source reads
                sum_mode=int(rng.choice([1, 1, 0])), quant_mode=int(rng.choice([1, 1, 0])))
    for k, v in opts.items():
        idx.set_option(k, v)
    tables = float_tables(rng, nq, ma, M, scale=float(rng.choice([0.2, 1.0])))
    if rng.integers(0, 2):
        tables = np.round(tables * 2) / 2                  # tie-heavy
    if rng.integers(0, 4) == 0:                            # a few negative entries (BLAS-expansion tables): clamped in place
        tables = np.where(rng.random(tables.shape) < 0.01, -np.float32(0.02) * tables, tables).astype(np.float32)
    res = idx.query_scan(assign, tables.copy(), R)
    for q in range(nq):
        want = po.query_scan(M, parts, labels, keep, assign[q], tables[q].copy(), R, quant_mode=opts["quant_mode"], sum_mode=opts["sum_mode"])
        if want["rc"] != 0:                                # the reference would exit: reported as status
            assert res["status"][q] == 1, (seed, q, opts)
            continue
        assert res["status"][q] == 0
        assert heaps_equal(res["heaps"][q], (want["keys"], want["values"])), (seed, q, sizes, R, keep, ma, opts)
    idx.close()
