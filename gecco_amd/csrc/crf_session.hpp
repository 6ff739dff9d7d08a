// Batch driver: host buffers in, host buffers out, over one or several devices.
//
// This is the layer `gecco run` reaches through ClusterCRF.predict_probabilities
// (gecco/crf/__init__.py:244-258: one loop iteration per contig, nothing shared between contigs):
// a batch of contigs is cut into chunks at contig boundaries, the chunks are dealt to the devices
// longest-first by gene count (per-GPU contig queues, no collective), and every
// device runs its queue through a ring of lanes.  A lane owns a stream, a reusable plan and device
// buffers; chunk k+1 is uploaded while chunk k computes and chunk k-1 is downloaded.  Nothing is
// allocated once a session has seen its largest chunk.
#pragma once
#include <cstdint>

#include "crf_plan.hpp"

namespace gecco {

struct BatchRequest {
    // batch (CSR over host memory; pinned memory from gecco_crf_host_alloc makes every copy asynchronous)
    const int32_t *contig_ptr = nullptr;
    int32_t n_contigs = 0;
    const int32_t *gene_ptr = nullptr, *attr_id = nullptr;
    // optional wire format: degree[i] = gene_ptr[i + 1] - gene_ptr[i] as bytes (every gene has at most 255 domains).  When
    // given, the degrees are what crosses PCIe (1 instead of 4 bytes per gene) and the row pointers are rebuilt on the
    // device; gene_ptr is then only read at chunk boundaries, on the host.
    const uint8_t *degree = nullptr;
    // optional wire format: the attribute indices as 16-bit words (a model with at most 65536 attributes; GECCO's has 2766)
    // instead of attr_id: 2 instead of 4 bytes per domain cross PCIe, widened on the device
    const uint16_t *attr_id16 = nullptr;
    int32_t window = 1, step = 1, label = 0, pad = 1;
    // what to compute, by output (null = not wanted)
    double *p_out = nullptr;        // [n_genes]      windowed marginals (row W)
    int8_t *y_out = nullptr;        // [n_genes]      Viterbi labels (row V)
    double *score_out = nullptr;    // [n_contigs]    Viterbi path scores
    double *marg_out = nullptr;     // [n_genes * L]  whole-contig marginals (row F)
    double *lognorm_out = nullptr;  // [n_contigs]
    // cluster calls (row R) on the windowed marginals, without moving them: rows and their count
    bool want_segments = false;
    const uint8_t *annotated = nullptr;  // [n_genes]
    SegParams seg;                       // carry is ignored (one grouper per contig); bio_ptr / bio_id are HOST arrays
                                         // over the batch's genes here (antismash criterion)
    int32_t *seg_out = nullptr;     // [max_seg][4] (contig, number, first gene, last gene + 1), global indices
    int32_t max_seg = 0;
    int32_t *n_seg = nullptr;
    double *seg_p_out = nullptr;    // [max_seg_genes] probabilities of the genes of the rows, row after row
    int64_t max_seg_genes = 0;
    int64_t *seg_off_out = nullptr; // [max_seg + 1] offsets of the rows in seg_p_out
};

struct SessionStats {
    int32_t n_chunks = 0, n_devices = 0;
    int64_t h2d_bytes = 0, d2h_bytes = 0;
    double host_plan_seconds = 0.0;  // time the submitting thread spent building chunk layouts
    double wall_seconds = 0.0;
    double host_issue_seconds = 0.0;  // time the submitting threads spent issuing work: HIP API calls + chunk layouts (all devices)
    int32_t host_threads = 1;         // threads that issued the batch's work
    int32_t direct = 0;  // the batch took the direct path (one chunk, kernels on pinned host memory, no copy commands)
};

struct Session;
int session_create(const Model &m, const int32_t *devices, int32_t n_devices, Session **out);
void session_destroy(Session *s);
int session_run(Session &s, const BatchRequest &r);
void session_set_chunk_genes(Session &s, int32_t genes);
void session_set_direct_genes(Session &s, int32_t genes);
void session_set_reference_bits(Session &s, bool on);
SessionStats session_stats(const Session &s);

}  // namespace gecco
