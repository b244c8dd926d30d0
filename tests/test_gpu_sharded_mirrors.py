"""The drop-in mirrors as a multi-GPU job (north_star: "the parent-filter and reference-subtract
stages shard the BAM read stream across the GPUs ... and merge per-GPU counts with an RCCL reduce"): two processes, one
per rank, run the mini-trio discovery chain through the SAME functions a one-process run calls
(discovery/pipeline.py:69-612 of the reference).  Both ranks share the one GPU of the test box, the process group is
gloo with host-staged collectives -- RCCL itself needs one GPU per rank (unmeasured on hardware)."""
import json
import os
import socket

import numpy as np
import pytest

from conftest import GIAB, GOLDEN

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _rank(rank, world, port, tmp, q):
    try:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        os.environ["KDF_READER_PIPELINES"] = "2"                    # two BGZF ranges per rank: 2 x world ranges of each BAM
        from kmer_denovo_filter_amd.core.jellyfish_wrappers import _scan_parent_jellyfish
        from kmer_denovo_filter_amd.discovery.pipeline import (
            _extract_child_kmers_discovery, _filter_parents_discovery, _subtract_reference_kmers)
        from kmer_denovo_filter_amd.kmer_fasta import read_kmer_fasta_keys
        fa, n1 = _extract_child_kmers_discovery(os.path.join(GIAB, "HG002_child.bam"), None, 31, 3, 4, tmp)
        cand = np.sort(read_kmer_fasta_keys(fa, 31)[0])
        fa2, n2 = _subtract_reference_kmers(os.path.join(GIAB, "mini_ref.fa.k31.jf"), fa, tmp)
        dist.barrier()
        gone = not os.path.exists(fa)
        nonref = np.sort(read_kmer_fasta_keys(fa2, 31)[0])
        # VCF mode's parent scan on the same shards: dict k-mer -> count summed over the ranks
        scan = _scan_parent_jellyfish(os.path.join(GIAB, "HG004_mother.bam"), None, fa2, 31, os.path.join(tmp, f"scan{rank}"), 4)
        n3, fa3 = _filter_parents_discovery(os.path.join(GIAB, "HG004_mother.bam"), os.path.join(GIAB, "HG003_father.bam"),
                                            None, fa2, 31, 4, tmp, 0)
        pu = np.sort(read_kmer_fasta_keys(fa3, 31)[0])
        q.put(("ok", rank, (n1, n2, n3), cand, nonref, pu, gone, scan))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as ex:  # noqa: BLE001
        import traceback
        q.put(("err", rank, f"{ex}\n{traceback.format_exc()}"))


@pytest.mark.parametrize("world", [2, 3])
def test_discovery_chain_sharded_over_ranks(oracle, trio_reads, tmp_path, world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    for r in res:
        assert r[0] == "ok", r[2]
    m = json.load(open(os.path.join(GOLDEN, "example_output_discovery", "giab_discovery.metrics.json")))
    ref = oracle.read_fasta(os.path.join(GIAB, "mini_ref.fa"))
    rt = oracle.OracleTable(31).count_reads([s for _, s in ref])
    st = oracle.discovery_chain(trio_reads["child"], trio_reads["mother"], trio_reads["father"], rt, 31, 3, 0)
    # the mother's counts of the non-reference k-mers (count >= 1), from the oracle
    mo = oracle.OracleTable(31).load_filter(st["non_ref"][0], np.zeros_like(st["non_ref"][0])).count_reads_filtered(trio_reads["mother"])
    mlo, _, mcnt = mo.export_ge(1)
    from kmer_denovo_filter_amd import keys_to_kmers
    exp_scan = dict(zip(keys_to_kmers(mlo, None, 31), mcnt.tolist()))
    for _, rank, (n1, n2, n3), cand, nonref, pu, gone, scan in res:
        assert (n1, n2, n3) == (m["child_candidate_kmers"], m["non_ref_kmers"], m["proband_unique_kmers"]) == (51125, 6679, 630)
        np.testing.assert_array_equal(cand, st["candidates"][0])
        np.testing.assert_array_equal(nonref, st["non_ref"][0])
        np.testing.assert_array_equal(pu, st["proband_unique"][0])
        assert gone
        assert scan == exp_scan
