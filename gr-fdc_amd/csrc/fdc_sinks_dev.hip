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

namespace fdc {

__device__ __forceinline__ unsigned long long lanemask_lt()
{
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    return lane ? (~0ull >> (64 - lane)) : 0ull;
}

// ---------------------------------------------------------------- PowerActivationChannel
// One lane per channel, one wave per 64 channels.  Per super-chunk of 2048 blocks: (1) the two comparisons of measure_power()
// (PowerActivationChannel_impl.cc:286-306) for every block — they need the previous block's power only, not the state — as bit
// masks in LDS, loads of consecutive blocks independent of each other; (2) the work() loop (:146-170) over those bits.
constexpr int kPacSuper = 2048;

__global__ __launch_bounds__(64) void k_pac_decide(const float *__restrict__ power, int ncells, int nb, const PacGeom *__restrict__ geom,
                                                   PacState *__restrict__ st, int npac, float thr, int mb, int R, long long bc0,
                                                   long long now, SinkTask *__restrict__ tasks, SinkPdu *__restrict__ pdus,
                                                   const int64_t *__restrict__ task_base, const int64_t *__restrict__ pdu_base,
                                                   int32_t *__restrict__ ntask, int32_t *__restrict__ npdu, SinkOwner *__restrict__ owners)
{
    __shared__ unsigned upw[kPacSuper / 32][64], dnw[kPacSuper / 32][64];
    const int lane = threadIdx.x, wv = blockIdx.x, i = wv * 64 + lane;
    const bool valid = i < npac;
    const PacGeom g = geom[valid ? i : 0];
    PacState s = st[valid ? i : 0];
    SinkTask *const tl = tasks + task_base[wv];
    SinkPdu *const pl = pdus + pdu_base[wv];
    int tcur = 0, pcur = 0;                            // wave-uniform cursors of this wave's lists
    const int carried = s.tail;
    int q = carried, E = 0;                            // stream length so far, emitted prefix
    const unsigned long long lt = lanemask_lt();
    float prev = s.lastpower;
    for (int m0 = 0; m0 < nb; m0 += kPacSuper) {
        const int mc = nb - m0 < kPacSuper ? nb - m0 : kPacSuper;
        // (1) ratios
        for (int w0 = 0; w0 < mc; w0 += 32) {
            unsigned u = 0, d = 0;
            const int wn = mc - w0 < 32 ? mc - w0 : 32;
            float p[32];
#pragma unroll
            for (int k = 0; k < 32; k++) p[k] = (k < wn && valid) ? power[(size_t)(m0 + w0 + k) * ncells + g.cell] : 1.0f;
#pragma unroll
            for (int k = 0; k < 32; k++) {
                float pw = p[k];
                if (pw == 0.0f) pw = FLT_MIN;                               // :293-294
                if (k < wn) {
                    if (pw / prev >= thr) u |= 1u << k;                     // :296
                    if (prev / pw >= thr) d |= 1u << k;                     // :299
                    prev = pw;                                              // lastpower follows every block (:297, :300, :304)
                }
            }
            upw[w0 >> 5][lane] = u; dnw[w0 >> 5][lane] = d;
        }
        // (2) work() loop
        for (int mm = 0; mm < mc; mm++) {
            const int m = m0 + mm;
            const unsigned u = (upw[mm >> 5][lane] >> (mm & 31)) & 1u, d = (dnw[mm >> 5][lane] >> (mm & 31)) & 1u;
            const bool rise = valid && !s.active && u, fall = valid && s.active && d;
            const bool proc = valid && (s.active || rise);                  // this block is extracted
            const unsigned long long b1 = __ballot(proc), b2 = __ballot(rise);
            if (b1) {
                const int pos = tcur + __popcll(b1 & lt) + __popcll(b2 & lt);
                if (rise) {                                                 // activate(), :198-210: previous and current block
                    s.part = 0; s.count = 0; s.active = 1; s.phase = 0; s.id_at_act = s.finished; s.act_time = now;
                    tl[pos] = SinkTask{i, q, m, g.extract_start, g.win_off, g.cls};                      // slot m = the block before
                    const int ph1 = g.deltaphase % R;
                    tl[pos + 1] = SinkTask{i, q + 1, m + 1, g.extract_start, g.win_off + ph1 * g.width, g.cls};
                    q += 2; s.count = 2; s.phase = (ph1 + g.deltaphase) % R;
                } else if (proc) {                                          // process_channel(), :260-284
                    tl[pos] = SinkTask{i, q, m + 1, g.extract_start, g.win_off + s.phase * g.width, g.cls};
                    q += 1; s.count += 1; s.phase = (s.phase + g.deltaphase) % R;
                }
                tcur += __popcll(b1) + __popcll(b2);
            }
            const bool part = proc && !rise && !fall && (mb == 0 || (mb > 0 && s.count % mb == 0));      // :163-165
            const bool emit = fall || part;
            const unsigned long long be = __ballot(emit);
            if (be) {
                if (emit) {                                                 // emit_data(), :212-258
                    SinkPdu r{};
                    r.key = ((long long)m << 24) | i;
                    r.blockstart = bc0 + m - s.count; r.blockend = bc0 + m; r.act_time = s.act_time;
                    r.owner = i; r.q0 = E; r.q1 = q; r.len = g.out_len;
                    r.kind = 0; r.source = g.id; r.chan_id = s.id_at_act; r.fin = fall ? 1 : 0; r.part = s.part; r.has_part = 1;
                    r.vstart = g.extract_start; r.vend = g.extract_start + g.width; r.width = g.width;
                    pl[pcur + __popcll(be & lt)] = r;
                    E = q; s.part += 1;
                    if (fall) { s.active = 0; s.finished += 1; }            // deactivate(), :189-196
                }
                pcur += __popcll(be);
            }
        }
    }
    if (valid) {
        s.lastpower = prev;
        SinkOwner o{};
        o.len = g.out_len; o.carried = carried; o.emitted = E; o.total = q; o.prev_off = s.tail_off;
        owners[i] = o;
        st[i] = s;                                     // tail / tail_off follow in k_sink_layout
    }
    if (lane == 0) { ntask[wv] = tcur; npdu[wv] = pcur; }
}

hipError_t launch_pac_decide(const float *power, int ncells, int nb, const PacGeom *geom, PacState *st, int npac, float thr,
                             int maxblocks, int R, long long bc0, long long now, SinkTask *tasks, SinkPdu *pdus,
                             const int64_t *task_base, const int64_t *pdu_base, int32_t *ntask, int32_t *npdu,
                             SinkOwner *owners, hipStream_t s)
{
    if (npac <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_pac_decide, dim3((unsigned)((npac + 63) / 64)), dim3(64), 0, s, power, ncells, nb, geom, st, npac, thr,
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
    __syncthreads();
    // std::sort by descending ratio (:713); equal ratios keep their order here (rank = elements in front in a stable sort)
    for (int a = lane; a < nr; a += 64) {
        const float ra = rr[a];
        int rank = 0;
        for (int b = 0; b < nr; b++) { const float rb = rr[b]; rank += (rb > ra || (rb == ra && b < a)) ? 1 : 0; }
        spos[rank] = rpos[a];
    }
    __syncthreads();
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
            __syncthreads();
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

// ---------------------------------------------------------------- detection, phase 2
// match_active_channels() + activation (…vcm_impl.cc:741-841), extract_channels_in_segments_singlethread() (:306-337) and
// clear_inactive_channels() (:512-524) for one segment: a loop over the blocks, lanes = live channels (chunks of 64).
constexpr int kDetLds = (int)(sizeof(int) * kDetFields * kDetMaxCells + sizeof(long long) * kDetMaxCells + sizeof(int) * 4 * kDetMaxCells +
                              sizeof(int) * (kDetMaxCells / 2 + 2) + sizeof(int2) * (kDetMaxCells / 2 + 2));
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
                                                  int32_t *__restrict__ error)
{
    // LDS (dynamic, kDetLds bytes): the live list (DetCol columns), the stream bookkeeping of each live channel, the candidates
    extern __shared__ __attribute__((aligned(16))) unsigned char fdc_det_smem[];
    int (*L)[kDetMaxCells] = reinterpret_cast<int (*)[kDetMaxCells]>(fdc_det_smem);                 // [kDetFields][cap]
    long long *oP = reinterpret_cast<long long *>(fdc_det_smem + sizeof(int) * kDetFields * kDetMaxCells);   // where its carried blocks lie
    int *oE = reinterpret_cast<int *>(oP + kDetMaxCells), *oQ = oE + kDetMaxCells, *oC = oQ + kDetMaxCells;  // emitted, total, carried
    int *hit = oC + kDetMaxCells, *claimed = hit + kDetMaxCells;
    int2 *cd = reinterpret_cast<int2 *>(claimed + kDetMaxCells / 2 + 2);
    const int lane = threadIdx.x, sg = blockIdx.x;
    const DetGeom g = geom[sg];
    const int lst = dp.npac ? (dp.npac + 63) / 64 + sg : sg;               // list index of this segment
    SinkTask *const tl = tasks + task_base[lst];
    SinkPdu *const pl = pdus + pdu_base[lst];
    SinkOwner *const ow = owners + owner_base[sg];
    const int ow0 = (int)owner_base[sg];
    int32_t *const Lg = live_g + (size_t)sg * kDetFields * kDetMaxCells;
    int64_t *const Og = live_off_g + (size_t)sg * kDetMaxCells;
    const unsigned long long lt = lanemask_lt();
    int nlive = sst[sg].nlive, counter = sst[sg].counter;
    for (int c = lane; c < nlive; c += 64) {
        for (int f = 0; f < kDetFields; f++) L[f][c] = Lg[f * kDetMaxCells + c];
        oE[c] = 0; oQ[c] = oC[c] = L[DC_TAIL][c]; oP[c] = Og[c];
        L[DC_OWNER][c] = c;                                                 // owners of the channels alive at the start: 0 .. nlive-1
    }
    int nown = nlive, tcur = 0, pcur = 0, err = 0;
    __syncthreads();
    const int segname = (dp.variant == 1 && dp.segname0 >= 0 && dp.nseg == 1) ? dp.segname0 : g.id;
    for (int m = 0; m < nb; m++) {
        const int k = ncand[(size_t)sg * dp.nbmax + m];
        if (k == 0) {                                                       // :748-752
            for (int c = lane; c < nlive; c += 64) L[DC_INACT][c] += 1;
        } else {
            const int2 *cs = cand + cand_base[sg] + (size_t)m * g.cand_cap;
            for (int j = lane; j < k; j += 64) { cd[j] = cs[j]; claimed[j] = 0; }
            for (int c = lane; c < nlive; c += 64) hit[c] = 0;
            __syncthreads();
            // a candidate goes to the FIRST live channel it overlaps (the reference erases it from the list there, :757-766)
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
            __syncthreads();
            for (int c = lane; c < nlive; c += 64) L[DC_INACT][c] = hit[c] ? 0 : L[DC_INACT][c] + 1;
            // what is left becomes new channels, in candidate order (:785-841)
            for (int j0 = 0; j0 < k; j0 += 64) {
                const int j = j0 + lane;
                bool ok = false;
                int ew = 0, es = 0, cls = 0;
                int2 pc = make_int2(0, 0);
                if (j < k && !claimed[j]) {
                    pc = cd[j];
                    const int dw = pc.y - pc.x, mid = pc.x + dw / 2;
                    ew = pow2ceil_dev((int)ceil((double)dw * (1.0 + 2.0 * dp.puffer)));
                    if (ew <= dp.N) {
                        cls = 31 - __clz(ew);
                        if (win_off[cls] >= 0) {
                            ok = true;
                            es = mid - ew / 2;
                            int ee = mid + ew / 2;
                            if (es < 0) { es = 0; ee = ew; }
                            if (ee > dp.N) { ee = dp.N; es = dp.N - ew; }
                        }
                    }
                }
                const unsigned long long bo = __ballot(ok);
                const int nnew = __popcll(bo);
                if (nlive + nnew > kDetMaxCells) { err = 1; break; }
                if (ok) {
                    const int r = __popcll(bo & lt), c = nlive + r;
                    L[DC_ID][c] = counter + r; L[DC_DSTART][c] = pc.x; L[DC_DSTOP][c] = pc.y; L[DC_ESTART][c] = es; L[DC_CLS][c] = cls;
                    L[DC_COUNT][c] = 0; L[DC_PHASE][c] = 0; L[DC_PINC][c] = es % dp.R; L[DC_INACT][c] = -1; L[DC_PART][c] = 0;
                    L[DC_OWNER][c] = nown + r; L[DC_TAIL][c] = 0;
                    L[DC_TIME_LO][c] = (int)(unsigned)(now & 0xFFFFFFFFll); L[DC_TIME_HI][c] = (int)(now >> 32);
                    oE[c] = 0; oQ[c] = 0; oC[c] = 0; oP[c] = 0;
                }
                nlive += nnew; counter += nnew; nown += nnew;
            }
            if (err) break;
            __syncthreads();
        }
        // extract_channels_in_segments_singlethread(), :306-337.  The block counter of the dictionaries: vcm counts from 1
        // (…vcm_impl.cc:188), SegmentDetection from 0 (SegmentDetection_impl.cc:118)
        const long long bc = bc0 + m - (dp.variant == 1 ? 1 : 0);
        bool anyfin = false;
        for (int pass = 0; pass < (dp.variant == 1 ? 2 : 1); pass++)
            for (int c0 = 0; c0 < nlive; c0 += 64) {
                const int c = c0 + lane;
                const bool in = c < nlive;
                int inact = in ? L[DC_INACT][c] : 0;
                bool isnew = false, fin = false, proc = false;
                if (pass == 0) { isnew = in && inact < 0; fin = in && !isnew && inact > dp.delay; proc = in && !isnew && !fin; }
                const unsigned long long b1 = __ballot(isnew || proc), b2 = __ballot(isnew);
                int E = in ? oE[c] : 0, Q = in ? oQ[c] : 0;
                const int cls = in ? L[DC_CLS][c] : 0, ew = 1 << cls, es = in ? L[DC_ESTART][c] : 0;
                const int own = in ? L[DC_OWNER][c] : 0;
                if (b1) {
                    const int pos = tcur + __popcll(b1 & lt) + __popcll(b2 & lt);
                    const int wo = win_off[cls];
                    if (isnew) {                                            // process_channel_hist(), :399-403
                        const int pinc = L[DC_PINC][c];
                        tl[pos] = SinkTask{ow0 + own, Q, m, es, wo, cls};
                        tl[pos + 1] = SinkTask{ow0 + own, Q + 1, m + 1, es, wo + (pinc % dp.R) * ew, cls};
                        Q += 2; L[DC_COUNT][c] = 2; L[DC_PHASE][c] = (2 * pinc) % dp.R; L[DC_INACT][c] = 0; inact = 0;
                    } else if (proc) {                                      // process_channel(), :373-397
                        const int ph = L[DC_PHASE][c];
                        tl[pos] = SinkTask{ow0 + own, Q, m + 1, es, wo + ph * ew, cls};
                        Q += 1; L[DC_COUNT][c] += 1; L[DC_PHASE][c] = (ph + L[DC_PINC][c]) % dp.R;
                    }
                    tcur += __popcll(b1) + __popcll(b2);
                }
                // partial emission: inline behind the channel in the vcm block (:317-318), a pass of its own in SegmentDetection (:359-362)
                const bool pcheck = in && dp.maxblocks >= 0 && (dp.variant == 1 ? pass == 1 : true) && !fin;
                int ntx = 0;
                if (pcheck && Q - E >= dp.maxblocks) ntx = dp.maxblocks == 0 ? Q - E : dp.maxblocks;
                const bool part = ntx > 0;
                const unsigned long long bf = __ballot(fin), bp = __ballot(part);
                if (bf | bp) {
                    if (fin || part) {
                        SinkPdu r{};
                        const int idx = dp.variant == 1 ? (pass == 0 ? c : nlive + c) : 2 * c + (part ? 1 : 0);
                        r.key = ((long long)m << 24) | (1ll << 23) | ((long long)sg << 12) | idx;
                        const int cnt = L[DC_COUNT][c];
                        r.blockstart = bc - cnt; r.blockend = bc;
                        r.act_time = ((long long)L[DC_TIME_HI][c] << 32) | (unsigned)L[DC_TIME_LO][c];
                        r.owner = ow0 + own; r.q0 = E; r.q1 = fin ? Q : E + ntx; r.len = ew - ew / dp.R;
                        r.kind = 1; r.source = segname; r.chan_id = L[DC_ID][c]; r.fin = fin ? 1 : 0; r.part = L[DC_PART][c];
                        r.has_part = fin ? (r.part > 0) : 1;
                        r.vstart = es; r.vend = es + ew; r.width = ew;
                        pl[pcur + __popcll(bf & lt) + __popcll(bp & lt)] = r;    // a lane emits at most one of the two in a pass
                        E = r.q1;
                        if (part) L[DC_PART][c] += 1;
                    }
                    pcur += __popcll(bf) + __popcll(bp);
                    anyfin = anyfin || bf != 0;
                }
                if (in) { oE[c] = E; oQ[c] = Q; }
            }
        __syncthreads();
        if (anyfin) {                                                       // clear_inactive_channels(), :512-524
            int keep = 0;
            for (int c0 = 0; c0 < nlive; c0 += 64) {
                const int c = c0 + lane;
                const bool in = c < nlive;
                const bool gone = in && L[DC_INACT][c] > dp.delay;
                int v[kDetFields], e = 0, qq = 0, cc = 0;
                long long pp = 0;
                if (in) {
#pragma unroll
                    for (int f = 0; f < kDetFields; f++) v[f] = L[f][c];
                    e = oE[c]; qq = oQ[c]; cc = oC[c]; pp = oP[c];
                }
                if (gone) {
                    SinkOwner o{};
                    o.len = (1 << v[DC_CLS]) - (1 << v[DC_CLS]) / dp.R; o.carried = cc; o.emitted = e; o.total = qq; o.prev_off = pp;
                    ow[v[DC_OWNER]] = o;
                }
                const unsigned long long bk = __ballot(in && !gone);
                __syncthreads();
                if (in && !gone) {
                    const int d = keep + __popcll(bk & lt);
#pragma unroll
                    for (int f = 0; f < kDetFields; f++) L[f][d] = v[f];
                    oE[d] = e; oQ[d] = qq; oC[d] = cc; oP[d] = pp;
                }
                keep += __popcll(bk);
                __syncthreads();
            }
            nlive = keep;
        }
    }
    __syncthreads();
    for (int c = lane; c < nlive; c += 64) {
        SinkOwner o{};
        const int cls = L[DC_CLS][c];
        o.len = (1 << cls) - (1 << cls) / dp.R; o.carried = oC[c]; o.emitted = oE[c]; o.total = oQ[c]; o.prev_off = oP[c];
        ow[L[DC_OWNER][c]] = o;
        for (int f = 0; f < kDetFields; f++) Lg[f * kDetMaxCells + c] = L[f][c];
    }
    if (lane == 0) {
        sst[sg].nlive = nlive; sst[sg].counter = counter;
        ntask[lst] = tcur; npdu[lst] = pcur; nowner[sg] = nown;
        if (err) *error = 1;
    }
}

hipError_t init_sink_kernels();
hipError_t launch_det_track(const DetParams &dp, int nb, const DetGeom *geom, DetSegState *sst, int32_t *live, int64_t *live_off,
                            const int2 *cand, const int64_t *cand_base, const int32_t *ncand, const int32_t *win_off,
                            long long bc0, long long now, SinkTask *tasks, SinkPdu *pdus, const int64_t *task_base,
                            const int64_t *pdu_base, int32_t *ntask, int32_t *npdu, SinkOwner *owners, const int64_t *owner_base,
                            int32_t *nowner, int32_t *error, hipStream_t s)
{
    if (dp.nseg <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_det_track, dim3((unsigned)dp.nseg), dim3(64), kDetLds, s, dp, nb, geom, sst, live, live_off, cand, cand_base, ncand,
                       win_off, bc0, now, tasks, pdus, task_base, pdu_base, ntask, npdu, owners, owner_base, nowner, error);
    return hipGetLastError();
}

// ---------------------------------------------------------------- layout
// One workgroup.  Owner table = regions: [0, npac) and, per segment s, [owner_base[s], owner_base[s] + nowner[s]).
// Landing buffer of the call: the emitted prefixes of all streams one behind the other (what goes to the host), then the
// buffered rests.
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
    long long accA = 0, accB = 0;
    int total_owner = 0;
    for (int rg = 0; rg <= nseg; rg++) {
        const long long o0 = rg == 0 ? 0 : owner_base[rg - 1];
        const int cnt = rg == 0 ? npac : nowner[rg - 1];
        total_owner += cnt;
        for (int c0 = 0; c0 < cnt; c0 += 1024) {
            const int c = c0 + tid;
            long long a = 0, b = 0;
            if (c < cnt) { const SinkOwner o = owners[o0 + c]; a = (long long)o.emitted * o.len; b = (long long)(o.total - o.emitted) * o.len; }
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
    // tasks per width class
    int nt = 0;
    for (int l = 0; l < nlist; l++) {
        const int n = ntask[l];
        nt += n;
        const SinkTask *t = tasks + task_base[l];
        for (int k = tid; k < n; k += 1024) atomicAdd(&hist[t[k].cls], 1);
    }
    // emission records, compacted; payload offsets
    int np = 0;
    for (int l = 0; l < nlist; l++) {
        const int n = npdu[l];
        const SinkPdu *p = pdus + pdu_base[l];
        for (int k = tid; k < n; k += 1024) {
            SinkPdu r = p[k];
            r.off = owners[r.owner].a_off + (long long)r.q0 * r.len;
            pdus_out[np + k] = r;
        }
        np += n;
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
    if (k >= ntask[l]) return;
    const SinkTask t = tasks[task_base[l] + k];
    const SinkOwner o = owners[t.owner];
    ExtractTask e{};
    e.slot = t.slot; e.start = t.start; e.win_off = t.win_off;
    e.out_off = t.q < o.emitted ? o.a_off + (long long)t.q * o.len : o.b_off + (long long)(t.q - o.emitted) * o.len;
    sorted[sum->class_base[t.cls] + atomicAdd(&class_fill[t.cls], 1)] = e;
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
    for (long long i = threadIdx.x; i < n; i += 256) (i < nA ? da : db)[i] = src[i];
}

hipError_t launch_carry_copy(const SinkOwner *owners, int nowner_cap, const int64_t *owner_base, const int32_t *nowner, int npac, int nseg,
                             const SinkSummary *sum, const float2 *prev, float2 *cur, hipStream_t s)
{
    (void)sum;
    if (nowner_cap <= 0 || !prev) return hipSuccess;
    hipLaunchKernelGGL(k_carry_copy, dim3((unsigned)nowner_cap, (unsigned)(nseg + 1)), dim3(256), 0, s, owners, owner_base, nowner, npac,
                       prev, cur);
    return hipGetLastError();
}

hipError_t init_sink_kernels()
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k_det_track), hipFuncAttributeMaxDynamicSharedMemorySize, kDetLds);
}

}  // namespace fdc
