import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
GIAB = os.path.join(GOLDEN, "giab")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def giab():
    return GIAB


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def trio_reads(oracle):
    """samtools-fasta reads of the mini trio, via the ORACLE's independent BAM reader."""
    return {
        "child": oracle.samtools_fasta_reads(os.path.join(GIAB, "HG002_child.bam")),
        "mother": oracle.samtools_fasta_reads(os.path.join(GIAB, "HG004_mother.bam")),
        "father": oracle.samtools_fasta_reads(os.path.join(GIAB, "HG003_father.bam")),
    }
