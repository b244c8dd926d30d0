"""Mirror of the k-mer-facing pieces of the reference's ``kmer_denovo_filter.vcf.pipeline``."""
