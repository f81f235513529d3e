#!/bin/sh
# TEST INFRASTRUCTURE — build-time extraction of line ranges of the reference's own source
# text, for oracle/Makefile's `ref_float` target.  Nothing here is product code.
#
# Why ranges: db_query_4.cpp, query_common.hpp, quantizers.hpp, databases.hpp and
# distances.hpp pull in Cereal / cblas / OpenCV headers that this image lacks, so the files do
# not compile whole; but the functions ON the hot path (QuantizerMAX, scan_4, scanner_4,
# base_pq, base_db, fmanorm, compute_dists_single_simd_cg ...) use none of those libraries.
# The ranges below are cut out of the files WHERE THEY LIE under $REF into a temporary
# directory, compiled from there with the reference's flags, and the directory is deleted
# before this script returns: no reference text is written under the repo (oracle/_ref/
# travels to the GPU box and must hold binaries only).  No stand-in header, library or macro
# is involved — the only lines of the cited structs that are left out are base_pq's two
# cereal `save`/`load` member templates (quantizers.hpp:170-187), which the path never calls,
# and, of struct opq, the two cblas rotate_* overrides and its cereal templates (279-323): the
# .opq.data reader (quantizers.cpp:40-46) only needs the rotation member and setup_rotation.
#
# Each range carries the first 16 hex digits of the sha256 of its text: if the reference
# drifts by a byte the build stops instead of silently compiling something else.
#
# usage: ref_extract.sh <REF dir> <out dir>     (out dir = a fresh mktemp -d of the caller)
set -eu
REF=$1
OUT=$2

cut_range() {   # file first last sha16 outname
    f=$REF/$1
    sed -n "$2,$3p" "$f" > "$OUT/$5"
    got=$(sha256sum < "$OUT/$5" | cut -c1-16)
    if [ "$got" != "$4" ]; then
        echo "ref_extract: $1:$2-$3 has sha256 $got, expected $4 — the reference changed; re-audit the ranges" >&2
        exit 1
    fi
}

#         file              first last  sha256[:16]        -> include name            what it holds
cut_range quantizers.hpp      24  169  a9214d90e36ee5a2  x_quantizers_a.inc   # subv, set_bits_generic, multiple_set_bits_native/_4, prepare_multiple_set_bits, extract_subvectors, class base_pq up to setup_centroids
cut_range quantizers.hpp     188  246  d6059b96a918beb0  x_quantizers_b.inc   # base_pq: rotate_*, code_size, encode_vector, encode_multiple_vectors, closing brace
cut_range databases.hpp       34   63  4f2edebd56931659  x_base_db.inc        # struct base_db
cut_range query_common.hpp    21   56  ea6e00041631c15b  x_query_metrics.inc  # struct query_metrics + operator<<
cut_range query_common.hpp    59  143  1e9518372b542c42  x_scan_funcs.inc     # scan_4<NSQ>, scan_standard<T,NSQ>, scan_func, get_scan_func
cut_range db_query_4.cpp      22  310  7fd2688931dc6d6b  x_scanner_4.inc      # simd_scan_func, get_simd_scan_func_epi8, QuantizerMAX<T>, struct scanner_4
cut_range db_query.cpp        17   46  96fef1dcccb6bcf3  x_scanner_simple.inc # struct scanner_simple
cut_range distances.hpp       21   36  9064091343d9c5a7  x_distances_a.inc    # SIMD_FLOATS, subv<DSQ>, reduceadd
cut_range distances.hpp       60   92  1f5eb31e1388ec18  x_distances_b.inc    # fmanorm<BLOCKS,REM> (the AVX2 branch config.h selects), both overloads
cut_range distances.hpp      237  275  6c3a81d8301bbef9  x_distances_c.inc    # centroids_getter, base_centroids_getter
cut_range distances.hpp      294  311  5e1490bd959a644e  x_distances_d.inc    # compute_dists_single_simd_cg<DSQ>
cut_range databases.cpp       24   48  4ceddaef15d5b5c3  x_substract.inc      # substract_vectors, substract_vectors_from_unique
# N3 / N1 (round 5): the .pq.data / .opq.data reader and the selection half of find_k_neighbors
cut_range quantizers.hpp     248  277  ff18229a99e946df  x_opq_a.inc          # struct opq up to set_rotation (the two cblas rotate_* overrides and the cereal templates, 279-323, are left out: the readers never call them)
cut_range quantizers.hpp     324  324  aa1d1a63390229c7  x_opq_z.inc          # its closing brace
cut_range quantizers.cpp      16   46  288255bdaacf4e70  x_pq_files_a.inc     # read_from_fstream, read_pq_from_fstream, pq_from_data_file(name, pq&), opq_from_data_file<>
cut_range quantizers.cpp      48  103  97a3c88ebb41e778  x_pq_files_b.inc     # invalid_data_filename, pq_type, parse_data_filename, pq_from_data_file(name)
cut_range neighbors.cpp       15   28  9d06318706e284de  x_neighbors_heaps.inc # BLOCK_VECS / BLOCK_NEIGHS, add_candidates_heaps
cut_range databases.cpp       70   88  32e83d7d4a358e15  x_kmeans_update.inc   # the centroid-update loops of kmeans_fast_iterations_thread (its assignment half is find_k_neighbors: cblas)
# N4 (round 6): the half of compute_cross_dists_blas that is NOT cblas — the norms and the ||v||^2 + ||c||^2 matrix it hands to
# sgemm as C (beta = 1).  Each range stops right before the "// BLAS Call" block; the harness closes the function body.
cut_range distances.hpp       51   57  e6e5667a9932e9dc  x_norm_4.inc          # norm_4
cut_range distances.hpp      151  176  f6a7290f0f54b133  x_cross_norms.inc     # compute_cross_dists_blas<DSQ>: head, centroid norms, distance-matrix loop (without 178-183: alpha/beta + cblas_sgemm + closing brace)
cut_range distances.hpp      185  208  a7d682be7297aaa0  x_cross_norms_4.inc   # its <4> specialisation, likewise (without 210-215)
