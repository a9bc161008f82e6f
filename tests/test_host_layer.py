"""The C++ host layer (quickstep_amd/host): RelationalOperator / WorkOrder mirror driving the
C ABI.  The C++ tests under tests/cpp mirror the reference's own operator unit tests; pytest
builds (if needed) and runs them."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "bin")


def _ensure_built():
    if not all(os.path.exists(os.path.join(BIN, b)) for b in
               ("select_cpu_workorder_test", "hash_join_operator_test", "aggregation_operator_test",
                "lip_filter_operator_test", "compressed_block_operator_test", "host_logic_test",
                "sort_operator_test", "nullable_operator_test", "tpch_types_operator_test", "tpch_q3_plan_test", "work_order_runs_test",
                "partition_operator_test", "headline_operators_bench", "block_image_test", "partitioned_ranks_test",
                "libloopback_rccl.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "quickstep_amd", "host")], check=True)


def _run(name, *args, timeout=600):
    _ensure_built()
    r = subprocess.run([os.path.join(BIN, name), *args], capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0 and "[  PASSED  ]" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


def test_select_operator_cpu_workorder_plumbing():
    """BASELINE config 1 (CPU WorkOrder through Foreman/Worker, no GPU) at its full size: a 10 M-row INTEGER column."""
    out = _run("select_cpu_workorder_test", "10000000", "4")
    assert out.count("M rows/s") == 3


def test_host_logic_without_a_device():
    """Compression choice + predicate rewriting on codes (row by row against the comparison on values), partition
    scheme bookkeeping, work-order container order: pure host logic of the operator layer."""
    _run("host_logic_test")


def test_gpu_operators_refuse_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    _ensure_built()
    r = subprocess.run([os.path.join(BIN, "hash_join_operator_test")], capture_output=True, text=True)
    assert r.returncode == 2 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_hash_join_operator_unittest_mirror():
    _run("hash_join_operator_test")


@pytest.mark.gpu
def test_aggregation_operator_unittest_mirror():
    _run("aggregation_operator_test")


@pytest.mark.gpu
def test_lip_filter_deployments_through_the_operators():
    """LIP.test data (R even, S multiples of 3): BuildHash builds the filter, Select / Aggregation /
    HashJoin(semi) probe it — sync driver and Foreman/Worker, exact and identity-hash filters."""
    _run("lip_filter_operator_test")


@pytest.mark.gpu
def test_compressed_column_store_blocks_through_the_operators():
    """CompressedBlockBuilder's choice per attribute, predicates rewritten to code comparisons and scanned on the code
    stripes, values decoded on demand for projections and aggregates: same results as over plain blocks."""
    _run("compressed_block_operator_test")


@pytest.mark.gpu
def test_sort_run_generation_and_merge_operators():
    """SortRunGenerationOperator / SortMergeRunOperator mirrors (1 and 3 ORDER BY columns, ASC/DESC/mixed, top-k),
    synchronous driver and Foreman/Worker with runs streaming into the merge."""
    _run("sort_operator_test")


@pytest.mark.gpu
def test_nulls_through_the_operators():
    """Join.test's LEFT JOIN chains (NULL join keys), Select.test's aggregates over the test table's NULLs, semi / anti joins
    and selections over nullable attributes — synchronous driver and Foreman/Worker."""
    _run("nullable_operator_test")


@pytest.mark.gpu
def test_tpch_schema_types_through_the_operators():
    """CHAR(10) / CHAR(1) / DATE attributes of the reference's TPC-H schema in the predicates and group-by keys of Q1 and Q3,
    over plain and over dictionary-compressed blocks."""
    _run("tpch_types_operator_test")


@pytest.mark.gpu
def test_tpch_q3_as_one_query_plan():
    """benchmarks/tpch/queries/03.sql as one operator DAG under ForemanSingleNode / four workers: three Selects, two
    BuildHash / HashJoin pairs, Aggregation over a 16-byte key with an expression argument, Finalize, SortRunGeneration,
    SortMergeRun with LIMIT 10 — streaming edges and pipeline breakers included."""
    _run("tpch_q3_plan_test")


@pytest.mark.gpu
def test_work_orders_over_runs_of_blocks():
    """SelectOperator / HashJoinOperator::setBlocksPerWorkOrder: a work order per run of blocks gives the tuples of a work
    order per block (select: same sequence, one output block per run; join: same multiset), and shapes the run form does
    not cover are executed block by block inside the work order."""
    out = _run("work_order_runs_test")
    assert "per run of 64 blocks" in out


@pytest.mark.gpu
def test_partition_test_known_answers_through_the_operators():
    """Partition.test: the 4-way hash partition listing produced by a repartitioning Select into a
    PartitionAwareInsertDestination (K9 scatter), partitioned / broadcast / REPARTITIONED hash joins, partitioned aggregation;
    has_repartition and the destination kind must agree (an error, never ignored)."""
    _run("partition_operator_test")


@pytest.mark.gpu
@pytest.mark.parametrize("exclusive_probes,cover", [("0", "1"), ("1", "1"), ("0", "0")])
def test_headline_workload_through_the_operator_boundary(exclusive_probes, cover):
    """BASELINE configs 2 + 3 as BuildHash / HashJoin / Aggregation / FinalizeAggregation operators under ForemanSingleNode on
    4 MB blocks with work orders over runs of blocks, scaled to 1 M x 20 M + 60 M rows; every step's results are checked
    (join condition on every output row, COUNT / SUM(qty) exact, sums to 1e-6).  Also with probe work orders keeping the
    device to themselves (QSX_HOST_EXCLUSIVE_PROBES=1: the Foreman's two classes of work orders) and with the projecting
    probe reading head[] and the stripes instead of a covering array (QSX_JOIN_COVER=0)."""
    _ensure_built()
    env = dict(os.environ, QSX_HOST_EXCLUSIVE_PROBES=exclusive_probes, QSX_JOIN_COVER=cover)
    r = subprocess.run([os.path.join(BIN, "headline_operators_bench"), "1000000", "20000000", "60000000", "3", "1", "4", "16"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and '"checked": true' in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("blocks_per_work_order", ["1", "16"])
def test_headline_workload_over_compressed_lineitem_images(blocks_per_work_order):
    """The same step with lineitem as the reference's TPC-H DDL stores it (benchmarks/tpch/create.sql:69-121): 4 MB
    CompressedColumnStore block images in the reference's layout, sorted on l_shipdate, quantity / discount / tax as 1-byte
    dictionary codes with a dictionary per block, adopted in place — stripes at whatever byte offsets the layout gives them.
    The AggregationOperator's work orders (single blocks and runs of 16) all go through the factored kernels
    (csrc/agg_factored.hpp); every step's groups are checked like the plain store's."""
    import json
    _ensure_built()
    r = subprocess.run([os.path.join(BIN, "headline_operators_bench"), "1000000", "20000000", "60000000", "3", "1", "4", blocks_per_work_order, "1"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and '"checked": true' in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert "CompressedColumnStore" in line["lineitem_store"]
    work_orders = -(-line["aggregate_blocks"] // int(blocks_per_work_order))
    # (a single block of 246 K rows stays below the factored kernels' default threshold of 256 Ki rows)
    assert line["factored_aggregation_launches_per_step"] == (float(work_orders) if blocks_per_work_order != "1" else 0.0)


@pytest.mark.gpu
def test_headline_workload_with_q1s_predicate_over_coded_shipdate():
    """lineitem_store = 2 of the operators bench: block images sorted on l_orderkey, l_shipdate a dictionary-coded attribute
    (2-byte codes, per-block dictionaries) and TPC-H Q1's l_shipdate <= DATE inside the AggregationOperator.  The state over
    code stripes leaves the predicate to the scans on codes (one qsx_select_codes_blocks per work order, the comparison
    rewritten on every block's own dictionary, stripes at whatever addresses the images give them) and keeps factoring its
    aggregates; groups checked against the rows that pass."""
    import json
    _ensure_built()
    r = subprocess.run([os.path.join(BIN, "headline_operators_bench"), "1000000", "20000000", "60000000", "3", "1", "4", "16", "2"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and '"checked": true' in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert "l_shipdate" in line["lineitem_store"]
    # every work order through the factored kernels — but for a last one of a single short block, which stays below their row
    # threshold (256 Ki rows) and takes the decoding kernels behind the same filter
    work_orders = -(-line["aggregate_blocks"] // 16)
    assert work_orders - 1 <= line["factored_aggregation_launches_per_step"] <= work_orders


@pytest.mark.gpu
def test_reference_block_images_are_adopted_in_place():
    """Block images in the reference's layout ([int32 header length][StorageBlockHeader][{num_tuples, nulls_in_sort_column}]
    [null bitmaps][stripes at max_tuples x width]) copied to device memory as they are: Select / Aggregation / HashJoin over
    the adopted blocks equal the same operators over blocks loaded column by column (nullable and non-nullable relations)."""
    _run("block_image_test")


@pytest.mark.gpu
def test_sharding_from_the_operator_layer_two_to_eight_rank_processes():
    """One process per rank, each with its own StorageManager / QueryContext / ForemanSingleNode running the same plan; rank r
    owns the partitions p % world == r; PartitionExchangeOperator (qsx_exchange_counts + qsx_alltoallv) moves the tuples of
    foreign partitions, ExchangeAggregationStatesOperator (qsx_agg_allgather_merge / qsx_agg_reduce_scatter) merges partial
    states; the ranks share cuda:0 over the loopback stand-in for RCCL.  Partition.test's partitioned / broadcast /
    repartitioned joins and aggregations and the BASELINE config 4 shape: union over ranks = the single-process operators.
    World 2, 3 (h % P), 4 and 8 — eight Foremen, the partition count of the node this is built for (C4 there with P = 8, one
    exchange round, and P = 16, two)."""
    out = _run("partitioned_ranks_test", timeout=900)
    assert all(f"world {w}:" in out for w in (2, 3, 4, 8))
