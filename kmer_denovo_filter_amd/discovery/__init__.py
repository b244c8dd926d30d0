"""Mirror of the hot-path helpers of the reference's ``kmer_denovo_filter.discovery.pipeline``."""
