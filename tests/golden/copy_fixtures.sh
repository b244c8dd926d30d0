#!/bin/bash
# copies the reference's test DATA (no source files) into tests/golden/
set -e
ref=${1:-/root/reference}; here="$(cd "$(dirname "$0")" && pwd)"
mkdir -p "$here/giab" "$here/example_output_discovery" "$here/example_output"
cp "$ref"/tests/data/giab/{HG002_child.bam,HG002_child.bam.bai,HG003_father.bam,HG004_mother.bam,mini_ref.fa,mini_ref.fa.fai,mini_ref.fa.k31.jf,candidates.vcf.gz} "$here/giab/"
cp "$ref"/tests/example_output_discovery/giab_discovery.{bed,kmer_coverage.bedgraph,metrics.json,read_coverage.bed,summary.txt,sv.bedpe} "$here/example_output_discovery/"
cp "$ref"/tests/example_output/{metrics.json,summary.txt} "$here/example_output/"
