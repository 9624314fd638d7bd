"""GPU: ipx_group (include/ipx.h) -- several GPUs behind one extractor, through the C ABI.  On the one-GPU
test box two members share the GPU (the exchange then runs over peer copies), and a group of one member is
sent through the exchange with IPX_GROUP_FORCE_EXCHANGE=1 so that the RCCL calls themselves run."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch            # noqa: F401  (before libipx.so: the HIP runtime torch brings along must be the one the process keeps)

from tests import helpers
from tests.oracle_binding import Oracle, OracleError

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle(batch, names, fasta=None, **params):
    ora = Oracle(targets=names, genome_fasta=fasta, **params)
    err = None
    try:
        ora.process(batch)
    except OracleError as e:
        err = e
    tab = ora.table()
    ora.close()
    return tab, err


@pytest.mark.parametrize("members,genome", [(2, False), (3, False), (2, True)])
def test_group_of_members_sharing_the_gpu_equals_oracle(members, genome, oracle_lib):
    import intronprospector_amd as ipa
    from intronprospector_amd import synth
    params, fasta, gen2, fetch = {}, None, None, None
    if genome:
        batch, names, gen = synth.short_reads(80_000, seed=61, n_targets=4, target_len=2_800_000, n_sites=900, genome=True)
        fasta = os.path.join(tempfile.mkdtemp(), "g.fa")
        gen.write_fasta(fasta)
        params = {"junction_filter": 2}
        gen2 = (gen.names, [(t, m, len(s)) for (t, m), s in zip(gen.twobit(), gen.seqs)])
        fetch = lambda tid, s, e: bytes([gen.seqs[tid][s], gen.seqs[tid][s + 1], gen.seqs[tid][e - 2], gen.seqs[tid][e - 1]])
    else:
        batch, names, _ = synth.short_reads(80_000, seed=62, n_targets=4, target_len=2_800_000, n_sites=900)
    grp = ipa.JunctionsExtractorGroup([0] * members, targets=names, genome=gen2, **params)
    assert not grp.uses_rccl()
    n = len(batch["tid"])
    for a, b in ((0, n // 3), (n // 3, n)):                       # two pushes, each spread over the members
        grp.push(synth.slice_batch(batch, a, b))
    dev = grp.get_junctions()
    ref, err = _oracle(batch, names, fasta, **params)
    assert err is None and "error" not in dev
    helpers.compare_tables(dev, ref, fetch)
    assert dev["n_alignments"] == n
    # a second run on the same group (clear keeps targets, genome, splitters)
    grp.clear()
    grp.push(batch, base_read_index=0)
    helpers.compare_tables(grp.get_junctions(), ref, fetch)
    grp.close()


def test_group_numbers_sjN_by_a_merge_in_ranges(oracle_lib):
    """Enough junctions (>= 4096) for ipx_group_finalize to cut the merge of its members' creation orders into ranges
    of the order axis, a thread each: the run-wide sjN must be the oracle's, junction by junction."""
    import intronprospector_amd as ipa
    from intronprospector_amd import synth
    batch, names, _ = synth.short_reads(150_000, seed=66, n_targets=4, target_len=6_000_000, n_sites=9000)
    grp = ipa.JunctionsExtractorGroup([0, 0, 0], targets=names)
    n = len(batch["tid"])
    for a, b in ((0, n // 4), (n // 4, n)):
        grp.push(synth.slice_batch(batch, a, b))
    dev = grp.get_junctions()
    ref, err = _oracle(batch, names)
    assert err is None and "error" not in dev
    assert len(ref["ijunc"]) >= 4096
    helpers.compare_tables(dev, ref, None)
    assert np.array_equal(np.sort(np.asarray(dev["ijunc"]).astype(np.int64)), np.arange(len(ref["ijunc"])))
    grp.close()


def test_group_members_fed_from_device_memory(oracle_lib):
    """ipx_group_push_member: the caller shards, every member takes its share from (its own) device memory -- how a
    step of the group is timed on resident input"""
    import intronprospector_amd as ipa
    from intronprospector_amd import synth
    batch, names, _ = synth.short_reads(90_000, seed=63, n_targets=4, target_len=2_800_000, n_sites=900)
    keys = ("tid", "pos", "flag", "xs", "nh", "cigar_off", "n_cigar", "cigar")
    grp = ipa.JunctionsExtractorGroup([0, 0, 0], targets=names)
    n = len(batch["tid"])
    cuts = [0, n // 4, n // 2, n]
    keep = []
    for m in range(3):
        part = synth.slice_batch(batch, cuts[m], cuts[m + 1])
        arrs = [torch.from_numpy(part[k]).to("cuda:0") for k in keys]
        keep.append(arrs)
        grp.push_member_device(m, len(part["tid"]), len(part["cigar"]), [a.data_ptr() for a in arrs], cuts[m])
    torch.cuda.synchronize()
    dev = grp.get_junctions()
    ref, err = _oracle(batch, names)
    assert err is None and "error" not in dev
    helpers.compare_tables(dev, ref)
    assert dev["n_alignments"] == n
    grp.close()


def test_group_raises_the_earliest_error(oracle_lib):
    """junctions_extractor.cc:73-74,312-313: a 'P' operation in the second member's range, a spliced FUNMAP read
    in the first one's: the group reports what the reference's loop would have thrown first"""
    import intronprospector_amd as ipa
    from intronprospector_amd import synth
    batch, names, _ = synth.short_reads(40_000, seed=63, n_targets=4, target_len=2_800_000, n_sites=700)
    n = len(batch["tid"])
    plain = np.nonzero((batch["n_cigar"] == 3) & (batch["flag"] == 0) & (batch["nh"] == 1))[0]
    i0 = int(plain[(plain > 100) & (plain < n // 2 - 100)][3])
    i1 = int(plain[plain > n // 2 + 100][3])
    batch["flag"][i0] |= 0x4
    w = int(batch["cigar_off"][i1])
    batch["cigar"][w] = (batch["cigar"][w] & ~np.uint32(0xf)) | np.uint32(6)
    _, err = _oracle(batch, names)
    assert err is not None and err.index == i0
    grp = ipa.JunctionsExtractorGroup([0, 0], targets=names)
    grp.push(batch)
    dev = grp.get_junctions()
    grp.close()
    assert "error" in dev and str(dev["error"]) == err.msg and dev["error"].read_index == i0
    # only the later one left: now that is the run's error
    batch["flag"][i0] &= ~np.uint16(0x4)
    _, err = _oracle(batch, names)
    grp = ipa.JunctionsExtractorGroup([0, 0], targets=names)
    grp.push(batch)
    dev = grp.get_junctions()
    grp.close()
    assert err.index == i1 and str(dev["error"]) == err.msg and dev["error"].read_index == i1


def test_group_of_one_runs_the_exchange_over_rccl(oracle_lib):
    """the RCCL leg (ncclCommInitAll, grouped ncclSend / ncclRecv to itself) in its own process"""
    code = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
import torch            # the HIP runtime and RCCL torch brings along are the ones libipx binds to
import intronprospector_amd as ipa
from intronprospector_amd import synth
from tests import helpers
from tests.oracle_binding import Oracle
batch, names, _ = synth.short_reads(60_000, seed=64, n_targets=4, target_len=2_800_000, n_sites=800)
grp = ipa.JunctionsExtractorGroup([0], targets=names)
assert grp.uses_rccl()
grp.push(batch)
dev = grp.get_junctions()
ora = Oracle(targets=names); ora.process(batch); ref = ora.table(); ora.close()
helpers.compare_tables(dev, ref)
grp.close()
print("rccl-exchange-ok", len(ref["intron_start"]))
""" % ROOT
    env = dict(os.environ, IPX_GROUP_FORCE_EXCHANGE="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl-exchange-ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_cli_devices_option(oracle_cli):
    """intronProspector --devices=0,0: same bytes as the single-GPU command line"""
    from tests import bamio          # noqa: F401  (the SAM fixtures are text)
    exe = os.path.join(ROOT, "intronprospector_amd", "bin", "intronProspector")
    sam = os.path.join(ROOT, "tests", "golden", "input", "test_basic.sam")
    with tempfile.TemporaryDirectory() as d:
        outs = []
        for tag, extra in (("one", []), ("two", ["--devices=0,0"])):
            tsv, bed = os.path.join(d, tag + ".tsv"), os.path.join(d, tag + ".bed")
            r = subprocess.run([exe, "--min-confidence-score=0.0", "-c", tsv, "-j", bed] + extra + [sam],
                               capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr
            outs.append((open(tsv).read(), open(bed).read()))
        assert outs[0] == outs[1] and len(outs[0][0]) > 500
        exp = open(os.path.join(ROOT, "tests", "golden", "expected", "testBasic1.introns.tsv")).read()
        assert outs[1][0] == exp


def test_bench_multi_group_config5_downscaled():
    """bench.py --multi group: config 5's shares (member m = rank m's contiguous 12.5 chunks) through ipx_group with two
    members on the test GPU, down-scaled; the line carries conservation, global sjN, key order and the oracle parity
    record -- what the driver-comparable line of the product's own multi-GPU path must hold at any size"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--multi", "group", "--gpus", "2", "--chunk", "200000",
                        "--steps", "2", "--warmup", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=root)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    cfg = line["config"]
    assert cfg["multi"] == "group" and cfg["members"] == 2
    for c in (cfg["checks"], cfg["parity"]["checks"]):
        assert c["conservation"] and c["ijunc_is_a_permutation"] and c["rows_in_key_order"], c
    assert cfg["parity"]["equals_oracle"], cfg["parity"]
    assert cfg["alignments"] == 2 * 25 * 100_000 and line["value"] > 0
