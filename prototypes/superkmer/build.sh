#!/bin/bash
# builds the experiment's library next to its source (hipcc cross-compiles without a GPU)
cd "$(dirname "$0")/../.." && hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Ikmer_denovo_filter_amd/csrc -Iinclude \
    -o prototypes/superkmer/libskproto.so prototypes/superkmer/sk_proto.hip
