// Device engine of the stateful sinks: the work() loops of the three sink blocks as gfx950 kernels.
//
//   gr::FDC::PowerActivationChannel       lib/PowerActivationChannel_impl.cc:137-306
//   gr::FDC::activity_detection_channelizer_vcm   lib/activity_detection_channelizer_vcm_impl.cc:551-568, :617-841, :306-337
//   gr::FDC::SegmentDetection             lib/SegmentDetection_impl.cc:131-362 (the twin; differences marked `sd`)
//
// These loops are sequential over the blocks of a stream and tiny per block; what makes them worth a kernel is that nothing has to
// leave the device between the power sums and the extractions.  The parts that do not depend on the state run in parallel over
// the blocks (power ratios of a PowerActivationChannel; edge detection, sort and candidate selection of a segment), the rest is a
// loop over the blocks with one lane per PowerActivationChannel / one wave per segment (lanes = live channels).
// Output of a call: extraction tasks and emission records that name blocks of per-channel streams (fdc_sinks_dev.h); the layout
// kernel places the streams in the landing buffer so that every PDU is one contiguous run and only emitted runs cross PCIe.
#include "fdc_sinks_dev.h"
#include <cfloat>
#include <cstdio>

namespace fdc {

__device__ __forceinline__ unsigned long long lanemask_lt()
{
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    return lane ? (~0ull >> (64 - lane)) : 0ull;
}

// A workgroup of the detection kernels is ONE wave: its LDS operations execute in program order, so making one lane's write
// visible to another lane needs no s_barrier — and above all not the wait for outstanding global stores that __syncthreads()
// implies (the task and record stores of a block would be waited for, ~1 us, at every list operation).  What is needed is that
// the compiler neither reorders nor caches LDS accesses across the point.
__device__ __forceinline__ void lds_sync()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------- PowerActivationChannel
// One wave per channel, lanes = 64 consecutive blocks.  The work() loop (PowerActivationChannel_impl.cc:146-170) is a
// two-state machine: block m toggles the state if (inactive and P[m] / P[m-1] >= thr) or (active and P[m-1] / P[m] >= thr)
// (measure_power(), :286-306; lastpower follows every block).  Both comparisons need the neighbouring powers only, so they
// are taken for 64 blocks at once (two ballots); the toggles are then found by walking the set bits of the mask the current
// state listens to (a scalar loop over the state CHANGES, not over the blocks), and everything else about a block follows in
// closed form from the toggle mask: whether it is extracted, its place in the channel's block stream, count and phase, the
// emissions (count % maxblocks, :163-165) and their part numbers.
__device__ __forceinline__ unsigned long long mask_lt(int m) { return m <= 0 ? 0ull : (m >= 64 ? ~0ull : ((1ull << m) - 1ull)); }
__device__ __forceinline__ int last_bit_below(unsigned long long msk, int m)      // highest set bit of msk below position m, or -1
{
    const unsigned long long x = msk & mask_lt(m);
    return x ? 63 - __clzll((long long)x) : -1;
}
// emissions of a run before its count reaches c (they happen at counts >= 3 that the maxblocks rule selects, :163-165)
__device__ __forceinline__ int emissions_before(int c, int mb)
{
    if (mb < 0 || c <= 3) return 0;
    if (mb == 0) return c - 3;
    return (c - 1) / mb - 2 / mb;                                           // multiples of mb in [3, c - 1]
}

__global__ __launch_bounds__(64) void k_pac_decide(const float *__restrict__ power, int ncells, int nb, const PacGeom *__restrict__ geom,
                                                   PacState *__restrict__ st, int npac, float thr, int mb, int R, long long bc0,
                                                   long long now, SinkTask *__restrict__ tasks, SinkPdu *__restrict__ pdus,
                                                   const int64_t *__restrict__ task_base, const int64_t *__restrict__ pdu_base,
                                                   int32_t *__restrict__ ntask, int32_t *__restrict__ npdu, SinkOwner *__restrict__ owners)
{
    const int lane = threadIdx.x, i = blockIdx.x;
    const PacGeom g = geom[i];
    const PacState s0 = st[i];
    SinkTask *const tl = tasks + task_base[i];
    SinkPdu *const pl = pdus + pdu_base[i];
    const int rm = R - 1, dph = g.deltaphase & rm;
    const float *const pcol = power + g.cell;
    // wave-uniform running state
    int active = s0.active, count = s0.count, finished = s0.finished, id_at = s0.id_at_act;
    long long act_time = s0.act_time;
    int q = s0.tail, E = 0, tcur = 0, pcur = 0;                              // stream length, emitted prefix, list cursors
    float last = s0.lastpower;
    for (int m0 = 0; m0 < nb; m0 += 64) {
        const int m = m0 + lane;
        const bool in = m < nb;
        float pw = in ? pcol[(size_t)m * ncells] : 1.0f;
        if (pw == 0.0f) pw = FLT_MIN;                                       // :293-294
        float pv = __shfl_up(pw, 1, 64);
        if (lane == 0) pv = last;
        const unsigned long long U = __ballot(in && pw / pv >= thr), D = __ballot(in && pv / pw >= thr);   // :296, :299
        const int nin = nb - m0 < 64 ? nb - m0 : 64;
        last = __shfl(pw, nin - 1, 64);
        // toggles: walk the changes
        unsigned long long T = 0;
        {
            int sa = active, pos = 0;
            while (pos < 64) {
                const unsigned long long c = (sa ? D : U) & ~mask_lt(pos);
                if (!c) break;
                const int t = __builtin_ctzll(c);
                T |= 1ull << t; sa ^= 1; pos = t + 1;
            }
        }
        const unsigned long long lt = mask_lt(lane);
        const bool before = (active ^ (__popcll(T & lt) & 1)) != 0;          // state when block m arrives
        const bool tog = (T >> lane) & 1ull;
        const bool rise = in && tog && !before, fall = in && tog && before, proc = in && (before || rise);
        const unsigned long long RISE = __ballot(rise), FALL = __ballot(fall), PROC = __ballot(proc);
        // the run this block belongs to: activated at block a of this chunk (rises and falls alternate, so that is the case
        // when the last rise at or below m is later than the last fall below m), or alive since before the chunk — then
        // nothing has toggled below m, and every block of the chunk so far was processed
        const int a = last_bit_below(RISE, lane + 1), lf = last_bit_below(FALL, lane);
        const bool mine = a >= 0 && a > lf;
        const int cnt = mine ? lane - a + 2 : count + lane + 1;             // count after this block is processed
        const int qpos = q + __popcll(PROC & lt) + __popcll(RISE & lt);      // stream position of this block's (first) extraction
        if (proc) {
            const int pos = tcur + __popcll(PROC & lt) + __popcll(RISE & lt);
            if (rise) {                                                      // activate(), :198-210: previous and current block
                tl[pos] = SinkTask{i, qpos, m, g.extract_start, g.win_off, g.cls};
                tl[pos + 1] = SinkTask{i, qpos + 1, m + 1, g.extract_start, g.win_off + dph * g.width, g.cls};
            } else {                                                         // process_channel(), :260-284: phase = (count - 1) deltaphase
                tl[pos] = SinkTask{i, qpos, m + 1, g.extract_start, g.win_off + (((cnt - 1) * dph) & rm) * g.width, g.cls};
            }
        }
        const bool part = proc && !rise && !fall && (mb == 0 || (mb > 0 && cnt % mb == 0));
        const bool emit = fall || part;
        const unsigned long long EM = __ballot(emit);
        if (emit) {                                                          // emit_data(), :212-258
            const int pe = last_bit_below(EM, lane);                         // the emission before this one flushed everything up to there
            const int q0 = pe >= 0 ? q + __popcll(PROC & mask_lt(pe + 1)) + __popcll(RISE & mask_lt(pe + 1)) : E;
            SinkPdu r;
            r.key = ((long long)m << 24) | i; r.act_time = mine ? now : act_time; r.off = 0;
            r.owner = i; r.q0 = q0; r.q1 = qpos + 1; r.count = cnt;
            r.chan_id = mine ? finished + __popcll(FALL & mask_lt(a)) : id_at;
            r.part = emissions_before(cnt, mb);
            r.flags = fall ? 1 : 0; r.vstart = g.extract_start;
            pl[pcur + __popcll(EM & lt)] = r;
        }
        // carry the chunk's end state on
        {
            const int nproc = __popcll(PROC) + __popcll(RISE);
            const int le_ = EM ? 63 - __clzll((long long)EM) : -1;            // last emission of the chunk
            if (le_ >= 0) E = q + __popcll(PROC & mask_lt(le_ + 1)) + __popcll(RISE & mask_lt(le_ + 1));
            q += nproc; tcur += nproc; pcur += __popcll(EM);
            const int lr = RISE ? 63 - __clzll((long long)RISE) : -1;
            const int act_end = active ^ (__popcll(T) & 1);
            if (act_end) {                                                   // (count / id of a finished run do not matter any more)
                if (lr >= 0) { count = nin - 1 - lr + 2; id_at = finished + __popcll(FALL & mask_lt(lr)); act_time = now; }
                else count += nin;
            }
            finished += __popcll(FALL);
            active = act_end;
        }
    }
    if (lane == 0) {
        PacState s = s0;
        s.lastpower = last; s.active = active; s.count = count; s.finished = finished; s.id_at_act = id_at; s.act_time = act_time;
        s.phase = (count * dph) & rm; s.part = 0;
        SinkOwner o{};
        o.len = g.out_len; o.cls = g.cls; o.carried = s0.tail; o.emitted = E; o.total = q; o.prev_off = s0.tail_off;
        owners[i] = o;
        st[i] = s;                                     // tail / tail_off follow in k_sink_layout
        ntask[i] = tcur; npdu[i] = pcur;
    }
}

hipError_t launch_pac_decide(const float *power, int ncells, int nb, const PacGeom *geom, PacState *st, int npac, float thr,
                             int maxblocks, int R, long long bc0, long long now, SinkTask *tasks, SinkPdu *pdus,
                             const int64_t *task_base, const int64_t *pdu_base, int32_t *ntask, int32_t *npdu,
                             SinkOwner *owners, hipStream_t s)
{
    if (npac <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_pac_decide, dim3((unsigned)npac), dim3(64), 0, s, power, ncells, nb, geom, st, npac, thr,
                       maxblocks, R, bc0, now, tasks, pdus, task_base, pdu_base, ntask, npdu, owners);
    return hipGetLastError();
}

// ---------------------------------------------------------------- detection, phase 1
// get_active_channels(), …vcm_impl.cc:694-739 (SegmentDetection_impl.cc:196-243): one wave per (block, segment).
__global__ __launch_bounds__(64) void k_det_cands(const float *__restrict__ power, int ncells, const DetGeom *__restrict__ geom, int dec,
                                                  float thr, int sd, int2 *__restrict__ cand, const int64_t *__restrict__ cand_base,
                                                  int32_t *__restrict__ ncand, int nbmax)
{
    __shared__ float rr[kDetMaxCells];
    __shared__ int rpos[kDetMaxCells], spos[kDetMaxCells], fpos[kDetMaxCells];
    __shared__ int2 acc[kDetMaxCells / 2 + 1];
    const int lane = threadIdx.x, m = blockIdx.x, sg = blockIdx.y;
    const DetGeom g = geom[sg];
    const float *P = power + (size_t)m * ncells + g.cell0;
    const float inv = 1.0f / thr;
    const unsigned long long lt = lanemask_lt();
    int nr = 0, nf = 0;
    for (int i0 = 1; i0 < g.ncell; i0 += 64) {
        const int i = i0 + lane;
        bool isr = false, isf = false;
        float pd = 0.f;
        if (i < g.ncell) {
            const float a = P[i - 1], b = P[i];
            // vcm guards a zero denominator (:703-706); SegmentDetection divides as is (volk_32f_x2_divide_32f, :206)
            pd = (!sd && a == 0.0f) ? b / FLT_MIN : b / a;
            isr = pd > thr;
            isf = pd < inv && !(sd && isr);            // SegmentDetection: if / else if (:209-210); vcm: two ifs (:708-709)
        }
        const unsigned long long br = __ballot(isr), bf = __ballot(isf);
        if (isr) { const int k = nr + __popcll(br & lt); rr[k] = pd; rpos[k] = (i - 1) * dec + g.start; }
        if (isf) fpos[nf + __popcll(bf & lt)] = i * dec + g.start;
        nr += __popcll(br); nf += __popcll(bf);
    }
    lds_sync();
    // std::sort by descending ratio (:713); equal ratios keep their order here (rank = elements in front in a stable sort)
    for (int a = lane; a < nr; a += 64) {
        const float ra = rr[a];
        int rank = 0;
        for (int b = 0; b < nr; b++) { const float rb = rr[b]; rank += (rb > ra || (rb == ra && b < a)) ? 1 : 0; }
        spos[rank] = rpos[a];
    }
    lds_sync();
    int nc = 0;
    for (int e = 0; e < nr; e++) {
        const int pos = spos[e];
        int ne = -1;                                                       // get_next_int(), :678-692: falling edges are in rising order
        for (int f0 = 0; f0 < nf; f0 += 64) {
            const unsigned long long mk = __ballot(f0 + lane < nf && fpos[f0 + lane] > pos);
            if (mk) { ne = fpos[f0 + __builtin_ctzll(mk)]; break; }
        }
        if (ne <= pos) continue;
        bool clash = false;                                                // :727-734
        for (int c0 = 0; c0 < nc; c0 += 64) {
            const bool hit = c0 + lane < nc && pos < acc[c0 + lane].y && ne >= acc[c0 + lane].x;
            if (__ballot(hit)) { clash = true; break; }
        }
        if (!clash) {
            if (lane == 0) acc[nc] = make_int2(pos, ne);
            nc++;
            lds_sync();
        }
    }
    int2 *out = cand + cand_base[sg] + (size_t)m * g.cand_cap;
    for (int c = lane; c < nc; c += 64) out[c] = acc[c];
    if (lane == 0) ncand[(size_t)sg * nbmax + m] = nc;
}

hipError_t launch_det_cands(const float *power, int ncells, int nb, const DetGeom *geom, int nseg, int dec, float thr, int sd,
                            int2 *cand, const int64_t *cand_base, int32_t *ncand, int nbmax, hipStream_t s)
{
    if (nseg <= 0 || nb <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_det_cands, dim3((unsigned)nb, (unsigned)nseg), dim3(64), 0, s, power, ncells, geom, dec, thr, sd, cand,
                       cand_base, ncand, nbmax);
    return hipGetLastError();
}

__device__ long long block_exscan(long long v, long long *tot, long long *sh /* [1024 / 64] */)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    long long x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const long long y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    __syncthreads();
    if (lane == 63) sh[wv] = x;
    __syncthreads();
    long long base = 0, all = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) { if (w < wv) base += sh[w]; all += sh[w]; }
    *tot = all;
    return base + x - v;
}

// ---------------------------------------------------------------- detection, phase 2
// match_active_channels() + activation (…vcm_impl.cc:741-841), extract_channels_in_segments_singlethread() (:306-337) and
// clear_inactive_channels() (:512-524) for one segment: a loop over the blocks of the call.
//
// This loop is a dependence chain (the list of live channels of block m decides what block m + 1 sees) run by ONE wave, so what
// counts is the number of instructions per block.  Three things keep it short:
//  * a live channel is extracted from EVERY block between its activation and its end, so no per-block extraction record is
//    written here: the channel's stream record says where its run starts, k_det_expand writes the tasks afterwards, in parallel;
//  * up to 64 live channels and 64 candidates (the normal case) a block is worked on from registers: lane c holds live channel c,
//    lane j candidate j; candidates x channels are compared as pairs in one step when they fit the wave;
//  * a workgroup is one wave: LDS needs no barrier (lds_sync), and nothing waits for stores.
// The list in LDS is the master copy (channels are appended and compacted there); count, inactivity, part and the stream
// counters live in registers until a compaction or the general form (more than 64 channels or candidates) needs them back.
constexpr int kDetStage = 32;        // blocks whose candidate lists are staged in LDS at a time (first 64 candidates of each)
constexpr int kDetLds = (int)(sizeof(int) * kDetFields * kDetMaxCells + sizeof(long long) * kDetMaxCells + sizeof(int) * 6 * kDetMaxCells +
                              sizeof(int) * (kDetMaxCells / 2 + 2) + sizeof(int2) * (kDetMaxCells / 2 + 2) +
                              sizeof(int2) * kDetStage * 64 + sizeof(int) * (kDetStage + 32));
static_assert(kDetLds <= 160 * 1024, "LDS budget of the detection tracker");
__device__ __forceinline__ int pow2ceil_dev(int k) { int p = 1; while (p < k) p <<= 1; return p; }

__global__ __launch_bounds__(64) void k_det_track(DetParams dp, int nb, const DetGeom *__restrict__ geom, DetSegState *__restrict__ sst,
                                                  int32_t *__restrict__ live_g, int64_t *__restrict__ live_off_g,
                                                  const int2 *__restrict__ cand, const int64_t *__restrict__ cand_base,
                                                  const int32_t *__restrict__ ncand, const int32_t *__restrict__ win_off, long long bc0,
                                                  long long now, SinkTask *__restrict__ tasks, SinkPdu *__restrict__ pdus,
                                                  const int64_t *__restrict__ task_base, const int64_t *__restrict__ pdu_base,
                                                  int32_t *__restrict__ ntask, int32_t *__restrict__ npdu, SinkOwner *__restrict__ owners,
                                                  const int64_t *__restrict__ owner_base, int32_t *__restrict__ nowner,
                                                  int32_t *__restrict__ error, long long *__restrict__ dbg)
{
#ifdef FDC_DET_STAMPS
    long long tS = 0, tM = 0, tX = 0, tC = 0, nG = 0, nC = 0, sK = 0, sL = 0;
    const long long t00 = __builtin_readcyclecounter();
#define DSTAMP(v) do { const long long _t = __builtin_readcyclecounter(); v += _t - tlast; tlast = _t; } while (0)
    long long tlast = t00;
#else
#define DSTAMP(v) do { } while (0)
#endif
    // LDS (dynamic, kDetLds bytes): the live list (DetCol columns), the stream bookkeeping of each live channel, the candidates
    extern __shared__ __attribute__((aligned(16))) unsigned char fdc_det_smem[];
    int (*L)[kDetMaxCells] = reinterpret_cast<int (*)[kDetMaxCells]>(fdc_det_smem);                 // [kDetFields][cap]
    long long *oP = reinterpret_cast<long long *>(fdc_det_smem + sizeof(int) * kDetFields * kDetMaxCells);   // where its carried blocks lie
    int *oE = reinterpret_cast<int *>(oP + kDetMaxCells), *oQ = oE + kDetMaxCells, *oC = oQ + kDetMaxCells;  // emitted, total, carried
    int *oS = oC + kDetMaxCells, *oH = oS + kDetMaxCells;                    // first spectrum slot of the call's run, its window phase
    int *hit = oH + kDetMaxCells, *claimed = hit + kDetMaxCells;
    int2 *cd = reinterpret_cast<int2 *>(claimed + kDetMaxCells / 2 + 2);
    int2 *stc = cd + kDetMaxCells / 2 + 2;                                   // [kDetStage][64]: staged candidates
    int *stk = reinterpret_cast<int *>(stc + kDetStage * 64);                // [kDetStage]: staged candidate counts
    int *wofs = stk + kDetStage;                                             // [32]: window table offsets per width class
    (void)tasks; (void)task_base; (void)ntask; (void)bc0;
    const int lane = threadIdx.x, sg = blockIdx.x;
    const DetGeom g = geom[sg];
    const int lst = dp.npac + sg;                                           // list index of this segment
    SinkPdu *const pl = pdus + pdu_base[lst];
    SinkOwner *const ow = owners + owner_base[sg];
    const int ow0 = (int)owner_base[sg];
    int32_t *const Lg = live_g + (size_t)sg * kDetFields * kDetMaxCells;
    int64_t *const Og = live_off_g + (size_t)sg * kDetMaxCells;
    const unsigned long long lt = lanemask_lt();
    const int rm = dp.R - 1;
    int nlive = sst[sg].nlive, counter = sst[sg].counter;
    for (int c = lane; c < nlive; c += 64) {
        for (int f = 0; f < kDetFields; f++) L[f][c] = Lg[f * kDetMaxCells + c];
        oE[c] = 0; oQ[c] = oC[c] = L[DC_TAIL][c]; oP[c] = Og[c];
        oS[c] = 1; oH[c] = L[DC_PHASE][c];                                  // goes on with the first block of the call (slot 1)
        L[DC_OWNER][c] = c;                                                 // owners of the channels alive at the start: 0 .. nlive-1
    }
    if (lane < 32) wofs[lane] = win_off[lane];
    int nown = nlive, pcur = 0, err = 0;
    lds_sync();
    const int32_t *const kc = ncand + (size_t)sg * dp.nbmax;
    const int2 *const cbase = cand + cand_base[sg];
    int rDS = 0, rDE = 0, rES = 0, rCLS = 0, rCNT = 0, rIN = 0, rPART = 0, rOWN = 0, rE = 0, rQ = 0, rID = 0, rTL = 0, rTH = 0;
    bool regs = false, pairs = false;
    int plw = 0, pc_ = 0, pj_ = 0, pDS = 0, pDE = 0;
    unsigned long long pstr = 0;
    auto load_static = [&]() {
        rDS = L[DC_DSTART][lane]; rDE = L[DC_DSTOP][lane]; rES = L[DC_ESTART][lane]; rCLS = L[DC_CLS][lane];
        rOWN = L[DC_OWNER][lane]; rID = L[DC_ID][lane]; rTL = L[DC_TIME_LO][lane]; rTH = L[DC_TIME_HI][lane];
    };
    auto load_regs = [&]() {
        if (lane < nlive) {
            load_static();
            rCNT = L[DC_COUNT][lane]; rIN = L[DC_INACT][lane]; rPART = L[DC_PART][lane]; rE = oE[lane]; rQ = oQ[lane];
        }
        regs = true;
    };
    auto spill_regs = [&]() {
        if (regs && lane < nlive) { L[DC_COUNT][lane] = rCNT; L[DC_INACT][lane] = rIN; L[DC_PART][lane] = rPART; oE[lane] = rE; oQ[lane] = rQ; }
        regs = false;
    };
    auto mkpdu = [&](int m, int idx, int own, int q0, int q1, int cnt, int id, int part, bool fin, int cls, int es, int tlo, int thi) {
        SinkPdu r;
        r.key = ((long long)m << 24) | (1ll << 23) | ((long long)sg << 12) | idx;
        r.act_time = ((long long)thi << 32) | (unsigned)tlo; r.off = 0;
        r.owner = ow0 + own; r.q0 = q0; r.q1 = q1; r.count = cnt; r.chan_id = id; r.part = part;
        r.flags = (fin ? 1 : 0) | (cls << 8) | (1 << 16); r.vstart = es;
        return r;
    };
    // geometry of a new channel from a candidate (:785-841); false = skipped (wider than the block, or no window table)
    auto new_geom = [&](int2 pc, int &es, int &cls) {
        const int dw = pc.y - pc.x, mid = pc.x + dw / 2;
        const int ew = pow2ceil_dev((int)ceil((double)dw * (1.0 + 2.0 * dp.puffer)));
        if (ew > dp.N) return false;
        cls = 31 - __clz(ew);
        if (wofs[cls] < 0) return false;
        es = mid - ew / 2;
        int ee = mid + ew / 2;
        if (es < 0) { es = 0; ee = ew; }
        if (ee > dp.N) { ee = dp.N; es = dp.N - ew; }
        return true;
    };
    auto append = [&](int c, int id, int2 pc, int es, int cls, int own, int m) {    // a new entry of the list in LDS
        L[DC_ID][c] = id; L[DC_DSTART][c] = pc.x; L[DC_DSTOP][c] = pc.y; L[DC_ESTART][c] = es; L[DC_CLS][c] = cls;
        L[DC_COUNT][c] = 0; L[DC_PHASE][c] = 0; L[DC_PINC][c] = es & rm; L[DC_INACT][c] = -1; L[DC_PART][c] = 0;
        L[DC_OWNER][c] = own; L[DC_TAIL][c] = 0;
        L[DC_TIME_LO][c] = (int)(unsigned)(now & 0xFFFFFFFFll); L[DC_TIME_HI][c] = (int)(now >> 32);
        oE[c] = 0; oQ[c] = 0; oC[c] = 0; oP[c] = 0;
        oS[c] = m; oH[c] = 0;                                               // process_channel_hist(): the block before (slot m) comes first
    };
    // One channel's share of extract_channels_in_segments_singlethread() (:306-337) in pass `pass` (SegmentDetection emits the
    // partial PDUs in a pass of its own, :359-362); every lane of the wave calls it.  Returns the lanes that finalised.
    auto step = [&](int m, int pass, int c, bool in, int &cnt, int &inact, int &part, int &E, int &Q, int cls, int es, int own, int id,
                    int tlo, int thi) -> unsigned long long {
        bool fin = false;
        if (pass == 0 && in) {
            if (inact < 0) { Q += 2; cnt = 2; inact = 0; }                  // process_channel_hist(), :399-403
            else if (inact > dp.delay) fin = true;                          // emit_channel(), :309
            else { Q += 1; cnt += 1; }                                      // process_channel(), :373-397
        }
        // partial emission: inline behind the channel in the vcm block (:317-318), a pass of its own in SegmentDetection
        const bool pcheck = in && dp.maxblocks >= 0 && (dp.variant == 1 ? pass == 1 : true) && !fin;
        int ntx = 0;
        if (pcheck && Q - E >= dp.maxblocks) ntx = dp.maxblocks == 0 ? Q - E : dp.maxblocks;
        const bool prt = ntx > 0;
        const unsigned long long bf = __ballot(fin), bp = __ballot(prt);
        if (bf | bp) {
            if (fin || prt) {
                const int idx = dp.variant == 1 ? (pass == 0 ? c : nlive + c) : 2 * c + (prt ? 1 : 0);
                const int q1 = fin ? Q : E + ntx;
                pl[pcur + __popcll(bf & lt) + __popcll(bp & lt)] =           // a lane emits at most one of the two in a pass
                    mkpdu(m, idx, own, E, q1, cnt, id, part, fin, cls, es, tlo, thi);
                E = q1;
                if (prt) part += 1;
            }
            pcur += __popcll(bf) + __popcll(bp);
        }
        return bf;
    };
    auto owner_record = [&](int c) {                                        // list entry c (everything in LDS) -> its stream record
        SinkOwner o{};
        const int cls = L[DC_CLS][c], w = 1 << cls;
        o.len = w - w / dp.R; o.cls = cls; o.carried = oC[c]; o.emitted = oE[c]; o.total = oQ[c]; o.prev_off = oP[c];
        o.slot0 = oS[c]; o.phase0 = oH[c]; o.pinc = L[DC_PINC][c]; o.estart = L[DC_ESTART][c]; o.win0 = wofs[cls];
        ow[L[DC_OWNER][c]] = o;
    };
    // clear_inactive_channels(), :512-524, on the list in LDS (everything spilled): finalised channels leave their stream record
    auto compact = [&]() {
        int keep = 0;
        for (int c0 = 0; c0 < nlive; c0 += 64) {
            const int c = c0 + lane;
            const bool in = c < nlive;
            const bool gone = in && L[DC_INACT][c] > dp.delay;
            int v[kDetFields], e = 0, qq = 0, cc = 0, ss = 0, hh = 0;
            long long pp = 0;
            if (gone) owner_record(c);
            if (in) {
#pragma unroll
                for (int f = 0; f < kDetFields; f++) v[f] = L[f][c];
                e = oE[c]; qq = oQ[c]; cc = oC[c]; pp = oP[c]; ss = oS[c]; hh = oH[c];
            }
            const unsigned long long bk = __ballot(in && !gone);
            lds_sync();
            if (in && !gone) {
                const int d = keep + __popcll(bk & lt);
#pragma unroll
                for (int f = 0; f < kDetFields; f++) L[f][d] = v[f];
                oE[d] = e; oQ[d] = qq; oC[d] = cc; oP[d] = pp; oS[d] = ss; oH[d] = hh;
            }
            keep += __popcll(bk);
            lds_sync();
        }
        nlive = keep;
    };
    for (int m = 0; m < nb; m++) {
        if ((m & (kDetStage - 1)) == 0) {                                   // candidate lists of the next kDetStage blocks -> LDS
            const int nst = nb - m < kDetStage ? nb - m : kDetStage;
            lds_sync();
            if (lane < nst) stk[lane] = kc[m + lane];
            const int cc = g.cand_cap < 64 ? g.cand_cap : 64;
            int2 tmp[kDetStage];                                            // all loads in flight before the first is used
#pragma unroll
            for (int t = 0; t < kDetStage; t++)
                tmp[t] = (t < nst && lane < cc) ? cbase[(size_t)(m + t) * g.cand_cap + lane] : make_int2(0, 0);
#pragma unroll
            for (int t = 0; t < kDetStage; t++) stc[t * 64 + lane] = tmp[t];
            lds_sync();
        }
        const int k = stk[m & (kDetStage - 1)];
        DSTAMP(tS);
#ifdef FDC_DET_STAMPS
        sK += k; sL += nlive;
#endif
        if (nlive <= 64 && k <= 64) {
            const int2 cme = stc[(m & (kDetStage - 1)) * 64 + lane];
            if (!regs) { load_regs(); pairs = false; }
            if (!pairs) {                                                   // pair layout of the matching: lane = (candidate pj, channel pc)
                plw = 0;
                while ((1 << plw) < nlive) plw++;
                pc_ = lane & ((1 << plw) - 1); pj_ = lane >> plw;
                pDS = __shfl(rDS, pc_, 64); pDE = __shfl(rDE, pc_, 64);
                // bits c, c + W, c + 2 W, ... of the pair mask: everything channel `lane` overlaps
                pstr = 0;
                for (int b_ = lane; b_ < 64; b_ += 1 << plw) pstr |= 1ull << b_;
                pairs = true;
            }
            // a candidate goes to the FIRST live channel it overlaps (the reference erases it from the list there, :757-766)
            unsigned long long hitm = 0, clmm = 0;                           // channels hit, candidates claimed
            if (k && nlive) {
                if ((k << plw) <= 64) {                                      // all pairs at once
                    const int cx = __shfl(cme.x, pj_, 64), cy = __shfl(cme.y, pj_, 64);
                    const unsigned long long ov = __ballot(pc_ < nlive && pj_ < k && cx < pDE && cy >= pDS);
                    const unsigned long long gm = plw == 6 ? ~0ull : (1ull << (1 << plw)) - 1ull;
                    const unsigned long long grp = lane < k ? ((ov >> (lane << plw)) & gm) : 0ull;     // channels candidate `lane` overlaps
                    clmm = __ballot(grp != 0);
                    if (!__ballot(grp & (grp - 1))) hitm = __ballot((ov & pstr) != 0);                   // nobody overlaps two: first = only
                    else
                        for (int jj = 0; jj < k; jj++) {
                            const unsigned long long gj = plw == 6 ? ov : ((ov >> (jj << plw)) & gm);
                            if (gj) hitm |= 1ull << __builtin_ctzll(gj);
                        }
                } else {
                    bool clm = false;
                    for (int c = 0; c < nlive; c++) {                       // channel c takes every candidate it overlaps
                        const int ds = __builtin_amdgcn_readlane(rDS, c), de = __builtin_amdgcn_readlane(rDE, c);
                        const bool ov = lane < k && !clm && cme.x < de && cme.y >= ds;
                        if (__ballot(ov)) { hitm |= 1ull << c; clm = clm || ov; }
                    }
                    clmm = __ballot(clm);
                }
            }
            {   // the quiet block — every candidate claimed, no channel past its delay — in a handful of instructions
                const int nin = (k != 0 && ((hitm >> lane) & 1ull)) ? 0 : rIN + 1;
                const bool inl = lane < nlive;
                if (!__ballot((lane < k && !((clmm >> lane) & 1ull)) || (inl && nin > dp.delay)) &&
                    (dp.maxblocks < 0 || !__ballot(inl && rQ + 1 - rE >= dp.maxblocks))) {
                    if (inl) { rIN = nin; rQ += 1; rCNT += 1; }
                    DSTAMP(tM);
                    continue;
                }
            }
            DSTAMP(tM);
            int nes = 0, ncls = 0;
            const bool ok = lane < k && !((clmm >> lane) & 1ull) && new_geom(cme, nes, ncls);   // the rest: new channels, in candidate order
            const unsigned long long bo = __ballot(ok);
            const int nnew = __popcll(bo);
            if (nlive + nnew <= 64) {
                if (lane < nlive) rIN = (k != 0 && ((hitm >> lane) & 1ull)) ? 0 : rIN + 1;      // :748-752, :768-771
                if (nnew) {
                    const int r = __popcll(bo & lt);
                    if (ok) append(nlive + r, counter + r, cme, nes, ncls, nown + r, m);
                    lds_sync();
                    if (lane >= nlive && lane < nlive + nnew) { load_static(); rCNT = 0; rIN = -1; rPART = 0; rE = 0; rQ = 0; }
                    nlive += nnew; counter += nnew; nown += nnew; pairs = false;
                }
                unsigned long long anyfin = 0;
                for (int pass = 0; pass < (dp.variant == 1 ? 2 : 1); pass++)
                    anyfin |= step(m, pass, lane, lane < nlive, rCNT, rIN, rPART, rE, rQ, rCLS, rES, rOWN, rID, rTL, rTH);
                DSTAMP(tX);
                if (anyfin) {
                    spill_regs();
                    lds_sync();
                    compact();
#ifdef FDC_DET_STAMPS
                    nC++;
#endif
                    DSTAMP(tC);
                }
                continue;
            }
        }
        // ---- general form: any number of channels and candidates, the list in LDS
#ifdef FDC_DET_STAMPS
        nG++;
#endif
        spill_regs();
        lds_sync();
        if (k == 0) {                                                       // :748-752
            for (int c = lane; c < nlive; c += 64) L[DC_INACT][c] += 1;
        } else {
            const int2 *cs = cbase + (size_t)m * g.cand_cap;
            for (int j = lane; j < k; j += 64) { cd[j] = cs[j]; claimed[j] = 0; }
            for (int c = lane; c < nlive; c += 64) hit[c] = 0;
            lds_sync();
            for (int j = 0; j < k; j++) {
                const int2 pc = cd[j];
                for (int c0 = 0; c0 < nlive; c0 += 64) {
                    const int c = c0 + lane;
                    const unsigned long long mk = __ballot(c < nlive && pc.x < L[DC_DSTOP][c] && pc.y >= L[DC_DSTART][c]);
                    if (mk) {
                        if (lane == 0) { hit[c0 + __builtin_ctzll(mk)] = 1; claimed[j] = 1; }
                        break;
                    }
                }
            }
            lds_sync();
            for (int c = lane; c < nlive; c += 64) L[DC_INACT][c] = hit[c] ? 0 : L[DC_INACT][c] + 1;
            for (int j0 = 0; j0 < k; j0 += 64) {                            // new channels, in candidate order (:785-841)
                const int j = j0 + lane;
                int es = 0, cls = 0;
                int2 pc = make_int2(0, 0);
                bool ok = false;
                if (j < k && !claimed[j]) { pc = cd[j]; ok = new_geom(pc, es, cls); }
                const unsigned long long bo = __ballot(ok);
                const int nnew = __popcll(bo);
                if (nlive + nnew > kDetMaxCells) { err = 1; break; }
                if (ok) { const int r = __popcll(bo & lt); append(nlive + r, counter + r, pc, es, cls, nown + r, m); }
                nlive += nnew; counter += nnew; nown += nnew;
            }
            if (err) break;
            lds_sync();
        }
        unsigned long long anyfin = 0;
        for (int pass = 0; pass < (dp.variant == 1 ? 2 : 1); pass++)
            for (int c0 = 0; c0 < nlive; c0 += 64) {
                const int c = c0 + lane, cc = c < kDetMaxCells ? c : kDetMaxCells - 1;
                anyfin |= step(m, pass, c, c < nlive, L[DC_COUNT][cc], L[DC_INACT][cc], L[DC_PART][cc], oE[cc], oQ[cc], L[DC_CLS][cc] & 31,
                               L[DC_ESTART][cc], L[DC_OWNER][cc], L[DC_ID][cc], L[DC_TIME_LO][cc], L[DC_TIME_HI][cc]);
            }
        lds_sync();
        if (anyfin) compact();
    }
    spill_regs();
    lds_sync();
    for (int c = lane; c < nlive; c += 64) {
        L[DC_PHASE][c] = (L[DC_COUNT][c] * L[DC_PINC][c]) & rm;             // window phase of the next block (:396)
        owner_record(c);
        for (int f = 0; f < kDetFields; f++) Lg[f * kDetMaxCells + c] = L[f][c];
    }
    if (lane == 0) {
        sst[sg].nlive = nlive; sst[sg].counter = counter;
        npdu[lst] = pcur; nowner[sg] = nown;
        if (err) *error = 1;
#ifdef FDC_DET_STAMPS
        if (dbg) {
            long long *o = dbg + sg * 16;
            o[0] = __builtin_readcyclecounter() - t00; o[1] = tS; o[2] = tM; o[3] = tX; o[4] = tC; o[5] = nG; o[6] = nC; o[7] = sK; o[8] = sL; o[9] = nb;
        }
#endif
    }
}

// The extractions of the detected channels of a call, written out from their stream records: block t of the run is spectrum
// slot slot0 + t, stream position carried + t, window phase (phase0 + t pinc) mod R (process_channel(), …vcm_impl.cc:373-397).
__global__ __launch_bounds__(1024) void k_det_expand(int npac, int R, const SinkOwner *__restrict__ owners, const int64_t *__restrict__ owner_base,
                                                     const int32_t *__restrict__ nowner, SinkTask *__restrict__ tasks,
                                                     const int64_t *__restrict__ task_base, int32_t *__restrict__ ntask)
{
    __shared__ long long sh[16];
    __shared__ int obase[1024];
    const int sg = blockIdx.x, tid = threadIdx.x, lst = npac + sg, rm = R - 1;
    const SinkOwner *ow = owners + owner_base[sg];
    const int ow0 = (int)owner_base[sg], cnt = nowner[sg];
    SinkTask *tl = tasks + task_base[lst];
    int acc = 0;
    for (int c0 = 0; c0 < cnt; c0 += 1024) {
        const int c = c0 + tid;
        const int n = c < cnt ? ow[c].total - ow[c].carried : 0;
        long long tot;
        obase[tid] = acc + (int)block_exscan((long long)n, &tot, sh);
        __syncthreads();
        const int lim = cnt - c0 < 1024 ? cnt - c0 : 1024, wv = tid >> 6, ln = tid & 63;
        for (int cc = wv; cc < lim; cc += 16) {
            const SinkOwner o = ow[c0 + cc];
            const int nn = o.total - o.carried, w = 1 << o.cls;
            for (int t = ln; t < nn; t += 64)
                tl[obase[cc] + t] = SinkTask{ow0 + c0 + cc, o.carried + t, o.slot0 + t, o.estart, o.win0 + ((o.phase0 + t * o.pinc) & rm) * w, o.cls};
        }
        acc += (int)tot;
        __syncthreads();
    }
    if (tid == 0) ntask[lst] = acc;
}

hipError_t launch_det_expand(int nseg, int npac, int R, SinkOwner *owners, const int64_t *owner_base, const int32_t *nowner, SinkTask *tasks,
                             const int64_t *task_base, int32_t *ntask, hipStream_t s)
{
    if (nseg <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_det_expand, dim3((unsigned)nseg), dim3(1024), 0, s, npac, R, owners, owner_base, nowner, tasks, task_base, ntask);
    return hipGetLastError();
}

hipError_t init_sink_kernels();
hipError_t launch_det_track(const DetParams &dp, int nb, const DetGeom *geom, DetSegState *sst, int32_t *live, int64_t *live_off,
                            const int2 *cand, const int64_t *cand_base, const int32_t *ncand, const int32_t *win_off,
                            long long bc0, long long now, SinkTask *tasks, SinkPdu *pdus, const int64_t *task_base,
                            const int64_t *pdu_base, int32_t *ntask, int32_t *npdu, SinkOwner *owners, const int64_t *owner_base,
                            int32_t *nowner, int32_t *error, hipStream_t s)
{
    if (dp.nseg <= 0) return hipSuccess;
    long long *dbg = nullptr;
#ifdef FDC_DET_STAMPS
    static long long *d_dbg = nullptr;
    if (!d_dbg) (void)hipMalloc(reinterpret_cast<void **>(&d_dbg), sizeof(long long) * 16 * 64);
    dbg = d_dbg;
#endif
    hipLaunchKernelGGL(k_det_track, dim3((unsigned)dp.nseg), dim3(64), kDetLds, s, dp, nb, geom, sst, live, live_off, cand, cand_base, ncand,
                       win_off, bc0, now, tasks, pdus, task_base, pdu_base, ntask, npdu, owners, owner_base, nowner, error, dbg);
#ifdef FDC_DET_STAMPS
    {
        long long h[32];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, dbg, sizeof h, hipMemcpyDeviceToHost);
        for (int g = 0; g < dp.nseg && g < 2; g++)
            std::fprintf(stderr, "[det_track seg %d] total %lld cyc; stage %lld match %lld step %lld compact %lld | general %lld compactions %lld sum k %lld sum live %lld of %lld blocks\n",
                         g, h[g * 16], h[g * 16 + 1], h[g * 16 + 2], h[g * 16 + 3], h[g * 16 + 4], h[g * 16 + 5], h[g * 16 + 6], h[g * 16 + 7], h[g * 16 + 8], h[g * 16 + 9]);
    }
#endif
    return hipGetLastError();
}

// ---------------------------------------------------------------- layout
// One workgroup.  Owner table = regions: [0, npac) and, per segment s, [owner_base[s], owner_base[s] + nowner[s]).
// Landing buffer of the call: the emitted prefixes of all streams one behind the other (what goes to the host), then the
// buffered rests.
__global__ __launch_bounds__(1024) void k_sink_layout(int nlist, const int64_t *__restrict__ task_base, const int64_t *__restrict__ pdu_base,
                                                      const int32_t *__restrict__ ntask, const int32_t *__restrict__ npdu,
                                                      const SinkTask *__restrict__ tasks, const SinkPdu *__restrict__ pdus,
                                                      SinkPdu *__restrict__ pdus_out, SinkOwner *__restrict__ owners, int npac, int nseg,
                                                      const int64_t *__restrict__ owner_base, const int32_t *__restrict__ nowner,
                                                      PacState *__restrict__ pst, DetSegState *__restrict__ sst, int32_t *__restrict__ live,
                                                      int64_t *__restrict__ live_off, SinkSummary *__restrict__ sum,
                                                      int32_t *__restrict__ class_fill, const int32_t *__restrict__ error)
{
    __shared__ long long sh[16];
    __shared__ int hist[32];
    const int tid = threadIdx.x;
    if (tid < 32) hist[tid] = 0;
    __syncthreads();
    long long accA = 0, accB = 0;
    int total_owner = 0;
    for (int rg = 0; rg <= nseg; rg++) {
        const long long o0 = rg == 0 ? 0 : owner_base[rg - 1];
        const int cnt = rg == 0 ? npac : nowner[rg - 1];
        total_owner += cnt;
        for (int c0 = 0; c0 < cnt; c0 += 1024) {
            const int c = c0 + tid;
            long long a = 0, b = 0;
            if (c < cnt) {
                const SinkOwner o = owners[o0 + c];
                a = (long long)o.emitted * o.len; b = (long long)(o.total - o.emitted) * o.len;
                if (o.total > o.carried) atomicAdd(&hist[o.cls & 31], o.total - o.carried);     // extractions of the call, per width class
            }
            long long ta, tb;
            const long long ea = block_exscan(a, &ta, sh);
            const long long eb = block_exscan(b, &tb, sh);
            if (c < cnt) { owners[o0 + c].a_off = accA + ea; owners[o0 + c].b_off = accB + eb; }
            accA += ta; accB += tb;
        }
    }
    const long long bstart = (accA + 1) & ~1ll;                            // 16-byte alignment of the second region
    __syncthreads();
    // the buffered rests: absolute offsets, and the persistent state that finds them again in the next call
    for (int rg = 0; rg <= nseg; rg++) {
        const long long o0 = rg == 0 ? 0 : owner_base[rg - 1];
        const int cnt = rg == 0 ? npac : nowner[rg - 1];
        for (int c = tid; c < cnt; c += 1024) {
            owners[o0 + c].b_off += bstart;
            if (rg == 0) { pst[c].tail = owners[c].total - owners[c].emitted; pst[c].tail_off = owners[c].b_off; }
        }
    }
    __syncthreads();
    for (int sg = 0; sg < nseg; sg++) {
        int32_t *Lg = live + (size_t)sg * kDetFields * kDetMaxCells;
        const int nl = sst[sg].nlive;
        for (int c = tid; c < nl; c += 1024) {
            const SinkOwner o = owners[owner_base[sg] + Lg[DC_OWNER * kDetMaxCells + c]];
            Lg[DC_TAIL * kDetMaxCells + c] = o.total - o.emitted;
            live_off[(size_t)sg * kDetMaxCells + c] = o.b_off;
        }
    }
    // emission records, compacted (list after list; the waves of the workgroup take lists in turn); payload offsets
    __shared__ int lbase[1024];
    int nt = 0, np = 0;
    for (int l0 = 0; l0 < nlist; l0 += 1024) {
        const int l = l0 + tid;
        const int n = l < nlist ? npdu[l] : 0;
        long long tot;
        const long long ex = block_exscan((long long)n, &tot, sh);
        lbase[tid] = np + (int)ex;
        long long tt;
        (void)block_exscan((long long)(l < nlist ? ntask[l] : 0), &tt, sh);
        nt += (int)tt;
        __syncthreads();
        const int wv = tid >> 6, ln = tid & 63, lim = nlist - l0 < 1024 ? nlist - l0 : 1024;
        for (int ll = wv; ll < lim; ll += 16) {
            const int nn = npdu[l0 + ll], b0 = lbase[ll];
            const SinkPdu *p = pdus + pdu_base[l0 + ll];
            for (int k = ln; k < nn; k += 64) {
                SinkPdu r = p[k];
                const SinkOwner o = owners[r.owner];
                r.off = o.a_off + (long long)r.q0 * o.len;
                pdus_out[b0 + k] = r;
            }
        }
        np += (int)tot;
        __syncthreads();
    }
    __syncthreads();
    if (tid < 32) { sum->class_cnt[tid] = hist[tid]; class_fill[tid] = 0; }
    if (tid == 0) {
        int acc = 0;
        for (int k = 0; k < 32; k++) { sum->class_base[k] = acc; acc += hist[k]; }
        sum->used_a = accA; sum->b_start = bstart; sum->used_total = bstart + accB;
        sum->npdu = np; sum->ntask = nt; sum->nowner = total_owner; sum->error = *error;
    }
}

hipError_t launch_sink_layout(int nlist, const int64_t *task_base, const int64_t *pdu_base, const int32_t *ntask, const int32_t *npdu,
                              const SinkTask *tasks, const SinkPdu *pdus, SinkPdu *pdus_out, SinkOwner *owners, int npac, int nseg,
                              const int64_t *owner_base, const int32_t *nowner, PacState *pst, DetSegState *sst, int32_t *live,
                              int64_t *live_off, SinkSummary *sum, int32_t *class_fill, const int32_t *error, hipStream_t s)
{
    hipLaunchKernelGGL(k_sink_layout, dim3(1), dim3(1024), 0, s, nlist, task_base, pdu_base, ntask, npdu, tasks, pdus, pdus_out, owners,
                       npac, nseg, owner_base, nowner, pst, sst, live, live_off, sum, class_fill, error);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_task_scatter(const int64_t *__restrict__ task_base, const int32_t *__restrict__ ntask,
                                                      const SinkTask *__restrict__ tasks, const SinkOwner *__restrict__ owners,
                                                      const SinkSummary *__restrict__ sum, int32_t *__restrict__ class_fill,
                                                      ExtractTask *__restrict__ sorted)
{
    const int l = blockIdx.y;
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool have = k < ntask[l];
    SinkTask t{};
    if (have) t = tasks[task_base[l] + k];
    // one atomic per wave and width class (a bank of equal channels has one class: a counter per task would serialise)
    int pos = 0;
    unsigned long long todo = __ballot(have);
    const unsigned long long lt = lanemask_lt();
    while (todo) {
        const int c0 = __builtin_amdgcn_readlane(t.cls, __builtin_ctzll(todo));
        const unsigned long long peers = __ballot(have && t.cls == c0) & todo;
        int base = 0;
        if ((threadIdx.x & 63) == __builtin_ctzll(peers)) base = atomicAdd(&class_fill[c0], __popcll(peers));
        base = __builtin_amdgcn_readlane(base, __builtin_ctzll(peers));
        if (have && t.cls == c0) pos = sum->class_base[c0] + base + __popcll(peers & lt);
        todo &= ~peers;
    }
    if (!have) return;
    const SinkOwner o = owners[t.owner];
    ExtractTask e{};
    e.slot = t.slot; e.start = t.start; e.win_off = t.win_off;
    e.out_off = t.q < o.emitted ? o.a_off + (long long)t.q * o.len : o.b_off + (long long)(t.q - o.emitted) * o.len;
    sorted[pos] = e;
}

hipError_t launch_task_scatter(int nlist, const int64_t *task_base, const int32_t *ntask, long long max_list, const SinkTask *tasks,
                               const SinkOwner *owners, const SinkSummary *sum, int32_t *class_fill, ExtractTask *sorted,
                               hipStream_t s)
{
    if (nlist <= 0 || max_list <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_task_scatter, dim3((unsigned)((max_list + 255) / 256), (unsigned)nlist), dim3(256), 0, s, task_base, ntask, tasks,
                       owners, sum, class_fill, sorted);
    return hipGetLastError();
}

// blocks buffered across the call boundary: from the previous landing buffer to their place in this one
constexpr int kCarrySplit = 16;     // workgroups per stream (a wide carrier buffers megabytes)
__global__ __launch_bounds__(256) void k_carry_copy(const SinkOwner *__restrict__ owners, const int64_t *__restrict__ owner_base,
                                                    const int32_t *__restrict__ nowner, int npac, const float2 *__restrict__ prev,
                                                    float2 *__restrict__ cur)
{
    const int rg = blockIdx.y, c = blockIdx.x;
    const int cnt = rg == 0 ? npac : nowner[rg - 1];
    if (c >= cnt) return;
    const SinkOwner o = owners[(rg == 0 ? 0 : owner_base[rg - 1]) + c];
    if (o.carried <= 0) return;
    const int ne = o.carried < o.emitted ? o.carried : o.emitted;         // carried blocks that went out in this call
    const float2 *src = prev + o.prev_off;
    float2 *da = cur + o.a_off, *db = cur + o.b_off - (long long)ne * o.len;
    const long long nA = (long long)ne * o.len, n = (long long)o.carried * o.len;
    for (long long i = (long long)blockIdx.z * 256 + threadIdx.x; i < n; i += 256 * kCarrySplit) (i < nA ? da : db)[i] = src[i];
}

hipError_t launch_carry_copy(const SinkOwner *owners, int nowner_cap, const int64_t *owner_base, const int32_t *nowner, int npac, int nseg,
                             const SinkSummary *sum, const float2 *prev, float2 *cur, hipStream_t s)
{
    (void)sum;
    if (nowner_cap <= 0 || !prev) return hipSuccess;
    hipLaunchKernelGGL(k_carry_copy, dim3((unsigned)nowner_cap, (unsigned)(nseg + 1), kCarrySplit), dim3(256), 0, s, owners, owner_base,
                       nowner, npac, prev, cur);
    return hipGetLastError();
}

hipError_t init_sink_kernels()
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k_det_track), hipFuncAttributeMaxDynamicSharedMemorySize, kDetLds);
}

}  // namespace fdc
