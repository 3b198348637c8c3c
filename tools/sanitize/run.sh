#!/bin/bash
# The host parser (ojph_t2.cpp, ojph_plan.cpp, ojph_pool.cpp, ht_tables.cpp -- no HIP in them) built with -fsanitize=address,undefined and
# run over a directory of damaged codestreams (main-header flips, data flips, cuts, scattered bytes; sources: random parameter
# sets written by the live reference and this library's Part-2 cases).  CPU only.      tools/sanitize/run.sh [sources] [seed]
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
W=/tmp/ojph_sanitize; mkdir -p $W; cd $W
for f in ojph_t2 ojph_plan ojph_pool ht_tables; do
  g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -I$ROOT/include -I$ROOT/openjph_amd/csrc -c $ROOT/openjph_amd/csrc/$f.cpp -o $f.o
done
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -I$ROOT/include $ROOT/tools/sanitize/parse_dir.cpp ojph_t2.o ojph_plan.o ojph_pool.o ht_tables.o -o parse_dir -lpthread
rm -rf corpus; python $ROOT/tools/sanitize/make_corpus.py $W/corpus ${1:-600} ${2:-5}
ASAN_OPTIONS=detect_leaks=1:allocator_may_return_null=1:max_allocation_size_mb=8000 UBSAN_OPTIONS=print_stacktrace=1 ./parse_dir corpus 2>&1 | tail -25
