// Device engine of the stateful sinks (fdc_sinks_dev.hip): records shared between the decision kernels and the host
// side (fdc_sinks.hip).  The work() loops of gr::FDC::PowerActivationChannel, activity_detection_channelizer_vcm and
// SegmentDetection run as kernels; the host only sizes buffers, launches the extractions and turns emission records into
// fdc_pdu.
#pragma once
#include "fdc_kernels.h"

namespace fdc {

constexpr int kDetMaxCells = 1024;      // power cells per detection segment the device engine takes (LDS-resident lists)

// One buffered block stream of a call: a PowerActivationChannel, or a detected channel that is alive during the call.
// The stream of a call = the blocks buffered before it (`carried`, lying in the previous call's landing buffer at
// prev_off) followed by the blocks extracted in it.  Every emission hands out a prefix of what is buffered
// (PowerActivationChannel_impl.cc:212-219 flushes everything, …vcm_impl.cc:454-470 the first maxblocks), so at the end of
// a call the first `emitted` blocks of the stream have gone out in PDUs and the rest stays buffered.
struct SinkOwner {
    int32_t len;                 // samples per block (output_len / outputsamples)
    int32_t cls;                 // log2 of the extraction width
    int32_t carried, emitted, total;
    // detected channels only: a channel is extracted from every block between its activation and its end, so the
    // extractions of a call are one run — blocks total - carried of them, the first from spectrum slot `slot0` with window
    // phase `phase0`, the phase advancing by `pinc` per block; k_det_expand writes the tasks out
    int32_t slot0, phase0, pinc, estart, win0, pad;
    int64_t prev_off;            // samples, in the previous call's landing buffer
    int64_t a_off, b_off;        // layout of this call (k_sink_layout): emitted prefix / buffered rest, samples in the landing buffer
};

// Extraction decided by a state machine: block `q` of stream `owner`
struct SinkTask { int32_t owner, q, slot, start, win_off, cls; };

// Emission record: blocks [q0, q1) of stream `owner`.  Kept small (the state machines write one per emission): what the
// host can work out itself — the dictionary's block numbers from the block index in `key` and `count`, the geometry of a
// PowerActivationChannel from `owner` — is not in it.
struct SinkPdu {
    int64_t key;                 // emission order inside a call: block << 40 | (PowerActivationChannel index, or 1 << 39 | segment << 28 |
                                 // pass << 27 | 2 * channel sequence + (partial ? 1 : 0))
    int64_t act_time;            // time() of the call that activated the channel (create_ID / get_ID_for_msg)
    int64_t off;                 // k_sink_layout: samples into the landing buffer
    int32_t owner, q0, q1, count;
    int32_t chan_id, part;
    int32_t flags;               // bit 0 finalized, bits 8-15 log2 width (detection), bit 16 kind (1 = detection)
    int32_t vstart;              // detection: first bin of the extraction
};

// PowerActivationChannel: geometry (constant) and state (lives on the device between calls)
struct PacGeom { int32_t cell, extract_start, width, cls, ovl, out_len, deltaphase, win_off, id, pad; };
struct PacState {
    float lastpower; int32_t active, count, phase, part, finished, id_at_act, tail;
    int64_t tail_off, act_time;
};

// Detection segment: geometry, and the list of live channels (structure of arrays, capacity kDetMaxCells per segment)
struct DetGeom { int32_t id, start, ncell, cell0, cand_cap, pad; };
constexpr int kDetFields = 14;   // int32 columns of the live list, see DetCol
enum DetCol { DC_ID, DC_DSTART, DC_DSTOP, DC_ESTART, DC_CLS, DC_COUNT, DC_PHASE, DC_PINC, DC_INACT, DC_PART, DC_OWNER,
              DC_TAIL, DC_TIME_LO, DC_TIME_HI };
struct DetSegState { int32_t nlive, counter; };

// What the host reads back after the decision + layout kernels of a call
struct SinkSummary {
    int64_t used_a, b_start, used_total;   // samples: emitted payloads [0, used_a) (copied to the host), buffered rest [b_start, used_total)
    int32_t class_cnt[32], class_base[32]; // extraction tasks per width class (log2 w), and where a class starts in the sorted array
    int32_t npdu, ntask, nowner, error;
    // what the launches that follow the layout really need (their grids were sized for the worst case before: ADVICE r03):
    int32_t max_list_tasks;                // the longest task list of the call (k_task_scatter's grid width)
    int32_t ncarry, max_region_owners;     // streams that carry blocks from the call before; the widest region of the owner table
    int32_t pad;
};

struct SinkLists {               // static partition of the task / record arrays into lists (one per wave of PACs, one per segment)
    int nlist;
    const int64_t *task_base, *pdu_base;   // device arrays [nlist + 1]
};

hipError_t init_sink_kernels();     // dynamic-LDS limit of the detection tracker (per device)

hipError_t launch_pac_decide(const float *power, int ncells, int nb, const PacGeom *geom, PacState *st, int npac, float thr,
                             int maxblocks, int R, long long bc0, long long now, SinkTask *tasks, SinkPdu *pdus,
                             const int64_t *task_base, const int64_t *pdu_base, int32_t *ntask, int32_t *npdu,
                             SinkOwner *owners, hipStream_t s);

// detection, phase 1 (every block on its own): edge detection, stable sort of the rising edges, candidate selection
// (…vcm_impl.cc:694-739 / SegmentDetection_impl.cc:196-243) -> cand[seg][block][cand_cap] (start, stop), ncand[seg][block]
hipError_t launch_det_cands(const float *power, int ncells, int nb, const DetGeom *geom, int nseg, int dec, float thr, int sd,
                            int2 *cand, const int64_t *cand_base /* [nseg] */, int32_t *ncand /* [nseg][nbmax] */, int nbmax,
                            hipStream_t s);

int det_track_staged(int nbmax, int max_cand_cap);   // < 0: the tracker's LDS tables do not fit (host engine)
struct DetParams {
    int N, R, dec, variant, maxblocks, delay, nseg, npac, nbmax, max_cand_cap;
    int mb_shift;                // log2(maxblocks) when it is a power of two >= 2, else -1
    double puffer;
    int segname0;                // SegmentDetection with one segment: the ID argument, else -1
    unsigned dec_magic;          // floor(2^32 / dec) + 1: n / dec = umulhi(n, dec_magic) for n * dec < 2^32 (set by launch_det_track)
};
// detection, phase 2 (one workgroup per segment; independent frequency regions of the segment in parallel waves): matching,
// activation, extraction bookkeeping, emissions (…vcm_impl.cc:741-841, :306-337; SegmentDetection_impl.cc:245-362).
// chs: scratch, one DetCh per entry of the owner table; live2: scratch of the size of `live`
struct DetCh { int32_t ds, de, es, cls, a, j, end, streak; };   // a channel's life in a call: detect range, extraction, activation block / candidate, end block
hipError_t launch_det_track(const DetParams &dp, int nb, const DetGeom *geom, DetSegState *sst, int32_t *live /* [nseg][kDetFields][kDetMaxCells] */,
                            int64_t *live_off /* [nseg][kDetMaxCells] */, const int2 *cand, const int64_t *cand_base,
                            const int32_t *ncand, const int32_t *win_off /* [log2 N + 1] */, long long now, SinkPdu *pdus,
                            const int64_t *pdu_base, int32_t *npdu, SinkOwner *owners, const int64_t *owner_base /* [nseg + 1], first = npac */,
                            int32_t *nowner /* [nseg] */, DetCh *chs, int32_t *live2,
                            int32_t *error /* set to 1 when a segment has more live channels than kDetMaxCells (read by k_sink_layout) */, hipStream_t s);

// extraction tasks of the detected channels of a call, from their stream records (one workgroup per segment)
hipError_t launch_det_expand(int nseg, int npac, int R, SinkOwner *owners, const int64_t *owner_base, const int32_t *nowner, SinkTask *tasks,
                             const int64_t *task_base, int32_t *ntask, hipStream_t s);

// layout of the landing buffer, class counts, emission records compacted to `pdus_out`, persistent tail offsets
hipError_t launch_sink_layout(int nlist, const int64_t *task_base, const int64_t *pdu_base, const int32_t *ntask, const int32_t *npdu,
                              const SinkTask *tasks, const SinkPdu *pdus, SinkPdu *pdus_out, SinkOwner *owners, int npac, int nseg,
                              const int64_t *owner_base, const int32_t *nowner, PacState *pst, DetSegState *sst, int32_t *live,
                              int64_t *live_off, SinkSummary *sum, int32_t *class_fill, const int32_t *error, hipStream_t s);

// summary and the first `eager` emission records into pinned host memory (kernel stores: no queueing behind a payload copy)
hipError_t launch_sink_publish(const SinkSummary *sum, const SinkPdu *pdus, SinkSummary *h_sum, SinkPdu *h_pdus, int eager, hipStream_t s);

// tasks -> ExtractTask grouped by width class (positions resolved through the owner table), order inside a class arbitrary
hipError_t launch_task_scatter(int nlist, const int64_t *task_base, const int32_t *ntask, long long max_list, const SinkTask *tasks,
                               const SinkOwner *owners, const SinkSummary *sum, int32_t *class_fill, ExtractTask *sorted,
                               hipStream_t s);

// buffered blocks of the previous call move to their place in this call's landing buffer
hipError_t launch_carry_copy(const SinkOwner *owners, int nowner_cap, const int64_t *owner_base, const int32_t *nowner, int npac, int nseg,
                             const SinkSummary *sum, const float2 *prev, float2 *cur, hipStream_t s);

}  // namespace fdc
