// Columnar host side of the path: table columns -> CSR batch, called clusters -> clusters.tsv rows.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

#include "../../include/gecco_crf.h"
#include "crf_model.hpp"

namespace gecco {

// std::vector that does not zero what it allocates (resize / sized construction default-initialise): the packer's arrays
// are tens of megabytes that its parallel passes write before anything reads them -- zeroing them first is a serial pass
// over fresh pages (2-4 ms per 0.4 M genes), whereas first touched inside the passes the page faults spread over the threads.
template <class T>
struct DefaultInitAlloc : std::allocator<T> {
    template <class U>
    struct rebind {
        using other = DefaultInitAlloc<U>;
    };
    using std::allocator<T>::allocator;
    template <class U>
    void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value) {
        ::new (static_cast<void *>(p)) U;
    }
    template <class U, class... Args>
    void construct(U *p, Args &&...args) {
        ::new (static_cast<void *>(p)) U(std::forward<Args>(args)...);
    }
};
template <class T>
using UVec = std::vector<T, DefaultInitAlloc<T>>;

struct Packed {
    int32_t n_genes = 0, n_contigs = 0;
    int64_t nnz = 0;
    int32_t n_duplicate_gene_ids = 0, n_unlisted_proteins = 0;
    bool pinned = false;
    // the CSR batch, in memory the batch driver copies from asynchronously (pinned when a device is present)
    char *block = nullptr;  // one allocation behind the four arrays
    int32_t *contig_ptr = nullptr, *gene_ptr = nullptr, *attr_id = nullptr;
    uint8_t *annotated = nullptr;
    int32_t *marker_ptr = nullptr, *marker_id = nullptr;  // [n_genes+1], [..]: marker domains per gene, or null
    UVec<int64_t> gene_row;   // [n_genes] gene-table row of every gene, or -1 - (its first feature row)
    UVec<int32_t> row_gene;   // [n_rows]  position (scoring order) of every feature row's gene
    UVec<int64_t> row_order;  // [n_rows]  feature rows by (gene position, domain_start), stable
    UVec<int64_t> row_ptr;    // [n_genes+1] offsets of every gene's rows in row_order
    ~Packed();
};

struct StrOut {
    std::vector<uint8_t> data;
    std::vector<int64_t> offsets;
};
struct ClusterRows {
    std::vector<int64_t> start, end;
    std::vector<double> average_p, max_p;
    StrOut sequence_id, cluster_id, proteins, domains;
};

int pack_columns(const Model &m, const gecco_crf_table_columns &t, Packed &out);
int cluster_rows(const Packed &pk, const gecco_crf_table_columns &t, const int64_t *gene_end, const int64_t *feat_end,
                 const int32_t *seg, int32_t n_seg, const double *seg_p, const int64_t *seg_off, ClusterRows &out);
double exact_mean(const double *v, int64_t n);
// out[i] = src[idx[i]] (idx in [0, n_src)), several host threads: the feature table's `cluster_probability` column
int gather_f64(const double *src, int64_t n_src, const int32_t *idx, int64_t n, double *out);
// What the output tables need to know of the scoring order (gene table rows given): are the genes' rows 0 .. n - 1 in order
// (the gene table can be handed back as it is), and do two genes of a contig share a start with their ends in decreasing
// order (the refiner's (start, end) order then differs from the scoring order, gecco/refine.py:190)?
int order_info(const Packed &pk, const int64_t *gene_start, const int64_t *gene_end, int64_t n_gene_rows, int32_t *rows_in_order,
               int32_t *refiner_order_differs);
int format_tsv(int64_t n_rows, int32_t n_cols, const int32_t *kinds, const void *const *data, const int64_t *const *offsets,
               const char *header, uint8_t **out, int64_t *out_len);

}  // namespace gecco
