"""Mirror of the reference's ``kmer_denovo_filter.core`` hot-path helpers."""
