// Device engine of the stateful sinks: the work() loops of the three sink blocks as gfx950 kernels.
//
//   gr::FDC::PowerActivationChannel       lib/PowerActivationChannel_impl.cc:137-306
//   gr::FDC::activity_detection_channelizer_vcm   lib/activity_detection_channelizer_vcm_impl.cc:551-568, :617-841, :306-337
//   gr::FDC::SegmentDetection             lib/SegmentDetection_impl.cc:131-362 (the twin; differences marked `sd`)
//
// These loops are sequential over the blocks of a stream and tiny per block; what makes them worth a kernel is that nothing has to
// leave the device between the power sums and the extractions.  The parts that do not depend on the state run in parallel over
// the blocks (power ratios of a PowerActivationChannel; edge detection, sort and candidate selection of a segment); the state machines
// themselves are turned so that lanes are BLOCKS: one wave per PowerActivationChannel (toggles found by walking the changes), and the
// detector channel by channel, slab by slab, independent frequency regions in parallel waves (k_det_track).
// Output of a call: extraction tasks and emission records that name blocks of per-channel streams (fdc_sinks_dev.h); the layout
// kernel places the streams in the landing buffer so that every PDU is one contiguous run and only emitted runs cross PCIe.
#include <climits>
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
            r.key = ((long long)m << 40) | i; r.act_time = mine ? now : act_time; r.off = 0;
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
// clear_inactive_channels() (:512-524) for one segment and a whole call.
//
// The reference walks the blocks and, inside a block, the list of live channels.  Walking it that way on a GPU is a chain of a
// thousand dependent, branchy steps run by one wave (measured: 1.3 us per block).  The same result comes out CHANNEL BY CHANNEL:
//   * a candidate of block m belongs to the first channel of the list that overlaps it (:757-766); the list is ordered by
//     activation, so if channels are handled in list order and each one marks the candidates it takes, a channel sees exactly
//     the candidates the reference would have left for it — in EVERY block of its life at once (lanes = blocks);
//   * its life ends at the first block where it has missed delay + 1 times in a row (:309) — a bit scan over the hit mask;
//   * everything else about it — its extractions (one run, k_det_expand), count, the partial emissions (:317-318, :454-470),
//     the final one — follows from (activation block, end block) in closed form;
//   * new channels are the candidates nobody took (:785-841): the sweep finds the first block that still has one once every
//     channel activated before that block has been handled.
// Channel by channel is still a chain (0.8 ms for the ~480 channels of a configs[4] step with one wave), but only channels that
// can compete for a candidate depend on each other, and only while both are alive.  So the workgroup (16 waves) takes the call
// in slabs of 64 blocks (lanes = the blocks of the slab).  Per slab it marks, on the segment's cell grid, every cell some
// candidate of the slab or some channel alive at its start covers; the maximal covered runs ("regions") cannot interact inside
// the slab — an overlap (:757, closed intervals) needs a common cell — and each is worked through by ONE wave, in the order
// above, the waves taking regions from a queue; the channels alive at the end of the slab, sorted by their place in the
// reference's list, are the input of the next one.  That pass only decides lives: (activation block, candidate, end block,
// misses in a row) per channel.  The sequence numbers the reference's order gives the new channels (block of activation, then
// candidate order inside the block) come from a prefix sum over the per-block masks of activating candidates afterwards, and
// then every channel's records — stream bookkeeping, emissions, its place in the next call's list — are written by whichever
// wave gets to it: nothing in them depends on another channel.
// Emission order inside a block is restored from the key: block, segment, pass (SegmentDetection sends its partial PDUs in a
// pass of their own, :359-362), position of the channel in the list (its sequence number: the list is ordered by activation).
constexpr int kDetWaves = 16;
constexpr int kDetStaged = 8;                        // candidates per block held in LDS (more than that: read from memory)
#ifndef FDC_DET_SLAB
#define FDC_DET_SLAB 64
#endif
constexpr int kDetSlab = FDC_DET_SLAB;               // blocks per slab (<= 64: lanes = blocks).  Measured on configs[4], payloads in HBM:
                                                     // 64 -> 1.00 ms per step, 32 -> 1.06, 16 -> 1.18 (shorter slabs split the regions
                                                     // further but pay the per-slab set-up more often)
__host__ __device__ constexpr size_t det_lds_bytes(int nb, int words)
{
    // activation masks + activation prefix per block of the call; per slab: candidate counts, staged candidates;
    // regions, two alive lists of five columns, small
    return (size_t)nb * (size_t)words * 8 + (size_t)nb * 4 + 64 * (4 + kDetStaged * 8) + (size_t)(kDetMaxCells / 2 + 1) * 8 +
           (size_t)kDetMaxCells * 40 + 16 * 8 + 16 * 8 + 32 * 4 + 64;
}
// value of a lane the whole wave agrees on (v_readlane: no trip through the LDS crossbar as for __shfl)
__device__ __forceinline__ int lane_val(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ unsigned long long lane_val(unsigned long long v, int l)
{
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
}
__device__ __forceinline__ int pow2ceil_dev(int k)
{
    if (k > (1 << 30)) return 0x7FFFFFFF;               // absurd flank puffer: wider than any block, the caller skips it
    return k <= 1 ? 1 : 1 << (32 - __clz(k - 1));
}

struct DetChanRegs {          // one channel while it is handled (wave-uniform)
    int id, ds, de, es, cls, cnt0, inact0, part0, buf0, a /* activation block, -1 = alive before the call */, own;
    int phase0, tlo, thi;
    long long prev_off;
};

template <int WORDS>        // 64-bit words of a block's taken-mask: 1, 2 or up to 8 (compile time: the word loops vanish for 1 and 2)
__global__ __launch_bounds__(64 * kDetWaves) void k_det_track(DetParams dp, int nb, const DetGeom *__restrict__ geom, DetSegState *__restrict__ sst,
                                                  int32_t *__restrict__ live_g, int64_t *__restrict__ live_off_g,
                                                  const int2 *__restrict__ cand, const int64_t *__restrict__ cand_base,
                                                  const int32_t *__restrict__ ncand, const int32_t *__restrict__ win_off,
                                                  long long now, SinkPdu *__restrict__ pdus, const int64_t *__restrict__ pdu_base,
                                                  int32_t *__restrict__ npdu, SinkOwner *__restrict__ owners,
                                                  const int64_t *__restrict__ owner_base, int32_t *__restrict__ nowner,
                                                  DetCh *__restrict__ chs_g, int32_t *__restrict__ live2_g, int32_t *__restrict__ error)
{
    constexpr int words = WORDS, staged = kDetStaged;
    extern __shared__ __attribute__((aligned(16))) unsigned char fdc_det_smem[];
    unsigned long long *NM = reinterpret_cast<unsigned long long *>(fdc_det_smem);       // [nb][words]: candidates that became channels
    int2 *CS = reinterpret_cast<int2 *>(NM + (size_t)nb * words);                        // [64][staged]: the first candidates of each block of the slab
    int2 *REG = CS + 64 * staged;                                                        // regions: (first cell, last cell)
    unsigned long long *COV = reinterpret_cast<unsigned long long *>(REG + (kDetMaxCells / 2 + 1));   // [16]: covered cells
    long long *sh = reinterpret_cast<long long *>(COV + 16);                             // [16]: block_exscan
    int *AL = reinterpret_cast<int *>(sh + 16);                                          // alive at the start of the slab, list order: 5 columns
    int *TL = AL + 5 * kDetMaxCells;                                                     // alive at its end, in the order the waves got there
    int *SV = AL, *SVS = TL;                                                             // afterwards: sequence numbers of the survivors, sorted positions
    int *NB = TL + 5 * kDetMaxCells;                                                     // [nb]: channels activated in earlier blocks
    int *KM = NB + nb;                                                                   // [64]: candidates per block of the slab
    int *wofs = KM + 64;                                                                 // [32]
    int *cnt = wofs + 32;                                                                // [0] regions [1] queue [2] new [3] records [4] survivors [5] alive (end of slab)
    enum { A_T = 0, A_KEY = kDetMaxCells, A_DS = 2 * kDetMaxCells, A_DE = 3 * kDetMaxCells, A_STK = 4 * kDetMaxCells };
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, sg = blockIdx.x;
    const DetGeom g = geom[sg];
    const int lst = dp.npac + sg;
    SinkPdu *const pl = pdus + pdu_base[lst];
    SinkOwner *const ow = owners + owner_base[sg];
    const int ow0 = (int)owner_base[sg];
    DetCh *const chs = chs_g + owner_base[sg];
    int32_t *const Lg = live_g + (size_t)sg * kDetFields * kDetMaxCells;
    int32_t *const L2 = live2_g + (size_t)sg * kDetFields * kDetMaxCells;
    int64_t *const Og = live_off_g + (size_t)sg * kDetMaxCells;
    const unsigned long long ltm = lanemask_lt();
    const int rm = dp.R - 1, mb = dp.maxblocks, sd = dp.variant == 1;
    const int32_t *const kc = ncand + (size_t)sg * dp.nbmax;
    const int2 *const cbase = cand + cand_base[sg];
    int m0 = 0;                                                                          // first block of the slab
    auto cand_of = [&](int m, int j) { return j < staged ? CS[(m - m0) * staged + j] : cbase[(size_t)m * g.cand_cap + j]; };
    auto cell_of = [&](int bin) {                                                        // (bin - start) / dec, clamped to the segment's cells
        const int n = bin - g.start;
        const int c = n <= 0 ? 0 : (dp.dec_magic ? (int)__umulhi((unsigned)n, dp.dec_magic) : n / dp.dec);
        return c >= g.ncell ? g.ncell - 1 : c;
    };
    auto cover = [&](int a, int b) {                                                     // cells a..b
        for (int w = a >> 6; w <= (b >> 6); w++) {
            const int lo = w == (a >> 6) ? (a & 63) : 0, hi = w == (b >> 6) ? (b & 63) : 63;
            const unsigned long long mk = (hi == 63 ? ~0ull : ((1ull << (hi + 1)) - 1ull)) & ~((1ull << lo) - 1ull);
            atomicOr(&COV[w], mk);
        }
    };

    for (int m = tid; m < nb; m += 64 * kDetWaves)
        for (int w = 0; w < words; w++) NM[(size_t)m * words + w] = 0;
    if (tid < 32) wofs[tid] = win_off[tid];
    if (tid < 8) cnt[tid] = 0;
    const int nlive0 = sst[sg].nlive, counter0 = sst[sg].counter;
    for (int i = tid; i < nlive0; i += 64 * kDetWaves) {                                 // the list the call starts from
        AL[A_T + i] = i; AL[A_KEY + i] = i; AL[A_DS + i] = Lg[DC_DSTART * kDetMaxCells + i]; AL[A_DE + i] = Lg[DC_DSTOP * kDetMaxCells + i];
        AL[A_STK + i] = Lg[DC_INACT * kDetMaxCells + i];
        DetCh h{}; h.a = -1; h.end = -1; chs[i] = h;
    }
    int nal = nlive0;
    __syncthreads();

    // a channel of the slab is either done (its end block is known) or goes on the list of the next slab
    auto settle = [&](int t, int key, int ds, int de, int end, int streak) {
        if (lane == 0) {
            if (end >= 0) chs[t].end = end;
            else {
                const int d = atomicAdd(&cnt[5], 1);
                if (d < kDetMaxCells) { TL[A_T + d] = t; TL[A_KEY + d] = key; TL[A_DS + d] = ds; TL[A_DE + d] = de; TL[A_STK + d] = streak; }
                else atomicExch(error, 1);                       // more live channels than power cells: the host refuses the batch (SinkSummary.error)
            }
        }
    };

#ifdef FDC_DET_STATS
    unsigned long long tacc[8] = {}, tlast = __builtin_readcyclecounter();
    int nregsum = 0;
    unsigned long long wacc[3] = {};                                 // this wave: region set-up, passes of alive channels, activations
    int wcnt[3] = {};                                                // regions, passes of alive channels, activations
#define FDC_WT(i, n) do { const unsigned long long tn = __builtin_readcyclecounter(); wacc[i] += tn - wlast; wlast = tn; wcnt[i] += (n); } while (0)
#define FDC_DT(i) do { const unsigned long long tn = __builtin_readcyclecounter(); tacc[i] += tn - tlast; tlast = tn; } while (0)
#else
#define FDC_WT(i, n) do { } while (0)
#define FDC_DT(i) do { } while (0)
#endif
    // the tables of a slab (candidate counts of its blocks, the first `staged` candidates of each) are requested a slab ahead, one entry
    // per thread (64 x staged = 512 <= the workgroup's threads), and only written to LDS here: their latency lies behind the slab before
    auto want_km = [&](int mb) { return (tid < kDetSlab && mb + tid < nb) ? kc[mb + tid] : 0; };
    auto want_cs = [&](int mb) {
        const int m = mb + tid / staged, j = tid % staged;
        return (tid < 64 * staged && m < nb && m < mb + kDetSlab && j < g.cand_cap) ? cbase[(size_t)m * g.cand_cap + j] : make_int2(0, 0);
    };
    static_assert(64 * kDetStaged <= 64 * kDetWaves, "one staged candidate per thread");
    int nkm = want_km(0);
    int2 ncs = want_cs(0);
    for (m0 = 0; m0 < nb; m0 += kDetSlab) {
        // ---- the slab's tables; cells covered by one of its candidates or by a channel alive at its start
        if (tid < 64) KM[tid] = nkm;
        if (tid < 64 * staged) CS[tid] = ncs;
        nkm = want_km(m0 + kDetSlab); ncs = want_cs(m0 + kDetSlab);
        if (tid < 16) COV[tid] = 0;
        if (tid == 0) { cnt[0] = 0; cnt[1] = 0; cnt[5] = 0; }
        __syncthreads();
        FDC_DT(0);
        {   // the staged candidates: a thread each; what a block has beyond them (rare): a wave per block
            const int q = tid / staged, j = tid % staged;
            if (tid < 64 * staged && j < KM[q]) { const int2 pc = CS[tid]; cover(cell_of(pc.x), cell_of(pc.y)); }
            for (int qq = wv; qq < kDetSlab && m0 + qq < nb; qq += kDetWaves) {
                const int k = KM[qq];
                for (int jj = staged + lane; jj < k; jj += 64) { const int2 pc = cbase[(size_t)(m0 + qq) * g.cand_cap + jj]; cover(cell_of(pc.x), cell_of(pc.y)); }
            }
        }
        for (int i = tid; i < nal; i += 64 * kDetWaves) cover(cell_of(AL[A_DS + i]), cell_of(AL[A_DE + i]));
        __syncthreads();
        FDC_DT(1);
        if (wv == 0) {
            // regions = maximal runs of covered cells: lane w looks at word w (ncell <= 1024 = 16 words)
            const unsigned long long c = lane < 16 ? COV[lane] : 0ull;
            const unsigned long long up = __shfl_up(c, 1, 64), dn = __shfl_down(c, 1, 64);
            const unsigned long long prev = (c << 1) | ((lane > 0 && lane < 16) ? up >> 63 : 0ull);
            const unsigned long long next = (c >> 1) | ((lane < 15) ? dn << 63 : 0ull);
            unsigned long long st = c & ~prev, en = c & ~next;              // first / last cell of a run
            int x = __popcll(st), base = x;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { const int y = __shfl_up(base, o, 64); if (lane >= o) base += y; }
            const int tot = __shfl(base, 15, 64);
            int is = base - x, ie = is + 0;
            // the k-th start and the k-th end belong together; runs may span words, so the ends are counted on their own
            int xe = __popcll(en), be = xe;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { const int y = __shfl_up(be, o, 64); if (lane >= o) be += y; }
            ie = be - xe;
            while (st) { REG[is++].x = 64 * lane + __builtin_ctzll(st); st &= st - 1; }
            while (en) { REG[ie++].y = 64 * lane + __builtin_ctzll(en); en &= en - 1; }
            if (lane == 0) cnt[0] = tot;
        }
        __syncthreads();
        const int nreg = cnt[0];
        FDC_DT(2);
#ifdef FDC_DET_STATS
        nregsum += nreg;
#endif
        for (;;) {
            int r = 0;
            if (lane == 0) r = atomicAdd(&cnt[1], 1);
            r = __builtin_amdgcn_readfirstlane(r);
            if (r >= nreg) break;
            // a candidate / channel belongs to the region its first bin lies in (bins of the region's cells: no division per test)
            const int b_lo = REG[r].x == 0 ? INT_MIN : g.start + REG[r].x * dp.dec;
            const int b_hi = REG[r].y >= g.ncell - 1 ? INT_MAX : g.start + (REG[r].y + 1) * dp.dec - 1;
            auto inreg = [&](int bin) { return bin >= b_lo && bin <= b_hi; };
            // This wave is the only one that looks at the region's candidates in this slab, and lane = block: what a block's
            // candidates are, which of them are the region's (RM), taken (cl) or have become channels (nm) stays in the lane's
            // registers for the whole region — a channel's pass over the slab is compares and ballots, no memory round trip.
#ifdef FDC_DET_STATS
            unsigned long long wlast = __builtin_readcyclecounter();
#endif
            const int m = m0 + lane;
            const int k = KM[lane];
            int2 cs[staged];
#pragma unroll
            for (int u = 0; u < staged; u++) cs[u] = CS[lane * staged + u];
            const bool deep = __ballot(k > staged) != 0;             // some block has more candidates than are staged (rare)
            int kmax = k;
            if (deep) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) kmax = max(kmax, __shfl_xor(kmax, o, 64));
            }
            unsigned long long RM[WORDS], cl[WORDS], nm[WORDS];
#pragma unroll
            for (int w = 0; w < WORDS; w++) { RM[w] = 0; cl[w] = 0; nm[w] = 0; }
#pragma unroll
            for (int u = 0; u < staged; u++) if (u < k && inreg(cs[u].x)) RM[0] |= 1ull << u;
            if (deep)
                for (int j = staged; j < k; j++)
                    if (inreg(cbase[(size_t)m * g.cand_cap + j].x)) {
#pragma unroll
                        for (int w = 0; w < WORDS; w++) if ((j >> 6) == w) RM[w] |= 1ull << (j & 63);
                    }
            // One channel over the blocks of the slab from `first` on: the candidates it takes (marked), the block of its end
            // (-1: it lives on), its misses in a row after the last block
            auto scan = [&](int ds, int de, int first, int inact0, int &end_out, int &streak_out) {
                const int f = first > m0 ? first - m0 : 0;     // first lane that takes part
                const bool valid = lane >= f && lane < kDetSlab && m < nb;
                unsigned long long ov[WORDS];
#pragma unroll
                for (int w = 0; w < WORDS; w++) ov[w] = 0;
#pragma unroll
                for (int u = 0; u < staged; u++)
                    if (u < k && cs[u].x < de && cs[u].y >= ds && !((cl[0] >> u) & 1ull)) ov[0] |= 1ull << u;
                if (deep)
                    for (int j = staged; j < kmax; j++)
                        if (j < k) {
                            const int2 pc = cbase[(size_t)m * g.cand_cap + j];
#pragma unroll
                            for (int w = 0; w < WORDS; w++)
                                if ((j >> 6) == w && pc.x < de && pc.y >= ds && !((cl[w] >> (j & 63)) & 1ull)) ov[w] |= 1ull << (j & 63);
                        }
                bool hit = false;
#pragma unroll
                for (int w = 0; w < WORDS; w++) hit = hit || ov[w] != 0;
                hit = hit && valid;
                const unsigned long long H = __ballot(hit);
                // misses in a row after block m's update (:748-752, :768-771)
                const unsigned long long hb = H & ltm;
                const int lh = hb ? 63 - __clzll((long long)hb) : -1;
                const int st = hit ? 0 : (lh >= 0 ? lane - lh : inact0 + (lane - f) + 1);
                const unsigned long long F = __ballot(valid && !hit && st > dp.delay);
                const int endlane = F ? __builtin_ctzll(F) : 64;
                if (hit && lane < endlane) {
#pragma unroll
                    for (int w = 0; w < WORDS; w++) cl[w] |= ov[w];
                }
                const int lastv = (nb - m0 < kDetSlab ? nb - m0 : kDetSlab) - 1;
                end_out = F ? m0 + endlane : -1;
                streak_out = lastv >= f ? lane_val(st, lastv) : inact0;
            };
            FDC_WT(0, 1);
            // the channels alive at the start of the slab, in list order
            for (int i0 = 0; i0 < nal; i0 += 64) {
                const int i = i0 + lane;
                unsigned long long B = __ballot(i < nal && inreg(AL[A_DS + (i < nal ? i : 0)]));
                while (B) {
                    const int ii = i0 + __builtin_ctzll(B);
                    B &= B - 1;
                    const int ds = AL[A_DS + ii], de = AL[A_DE + ii];
                    int end, streak;
                    scan(ds, de, m0, AL[A_STK + ii], end, streak);
                    settle(AL[A_T + ii], AL[A_KEY + ii], ds, de, end, streak);
                    FDC_WT(1, 1);
                }
            }
            // candidates of the region nobody took become channels (:785-841), block after block
            for (int ms = m0;;) {
                bool un = false;
                if (m < nb && m >= ms && lane < kDetSlab) {
#pragma unroll
                    for (int w = 0; w < WORDS; w++) un = un || (RM[w] & ~cl[w]) != 0;
                }
                const unsigned long long U = __ballot(un);
                if (!U) break;
                const int fl = __builtin_ctzll(U), mf = m0 + fl, kf = lane_val(k, fl);
                // the activating candidates of block mf; everything of the region in this block is now settled
#pragma unroll
                for (int w = 0; w < WORDS; w++) {
                    if (kf <= 64 * w) continue;                   // the block has no candidates in this word (wave-uniform)
                    const int j = 64 * w + lane;
                    bool ok = false;
                    const unsigned long long rmw = lane_val(RM[w], fl), clw = lane_val(cl[w], fl);
                    int2 pc = make_int2(0, 0);
                    int es = 0, cls = 0;
                    if (j < kf && ((rmw >> lane) & 1ull) && !((clw >> lane) & 1ull)) {
                        pc = cand_of(mf, j);
                        const int dw = pc.y - pc.x, mid = pc.x + dw / 2;
                        const int ew = pow2ceil_dev((int)ceil((double)dw * (1.0 + 2.0 * dp.puffer)));
                        if (ew <= dp.N) {                        // wider than the block: logged and skipped in the reference
                            cls = 31 - __clz(ew);
                            if (wofs[cls] >= 0) {
                                ok = true;
                                es = mid - ew / 2;
                                int ee = mid + ew / 2;
                                if (es < 0) { es = 0; ee = ew; }
                                if (ee > dp.N) { ee = dp.N; es = dp.N - ew; }
                            }
                        }
                    }
                    unsigned long long q = __ballot(ok);
                    if (lane == fl) { nm[w] |= q; cl[w] |= rmw; }
                    if (!q) continue;
                    int t0 = 0;
                    if (lane == 0) t0 = atomicAdd(&cnt[2], __popcll(q));
                    t0 = __builtin_amdgcn_readfirstlane(t0);
                    while (q) {
                        const int jl = __builtin_ctzll(q);
                        q &= q - 1;
                        const int cx = lane_val(pc.x, jl), cy = lane_val(pc.y, jl), ces = lane_val(es, jl), ccls = lane_val(cls, jl);
                        int end, streak;
                        scan(cx, cy, mf + 1, 0, end, streak);
                        const int t = nlive0 + t0;
                        t0++;
                        if (lane == 0) chs[t] = DetCh{cx, cy, ces, ccls, mf, 64 * w + jl, -1, 0};
                        settle(t, nlive0 + mf * g.cand_cap + 64 * w + jl, cx, cy, end, streak);
                        FDC_WT(2, 1);
                    }
                }
                ms = mf + 1;
            }
            // the activation masks of the region's blocks join the call's (ranks of the new channels, below)
#pragma unroll
            for (int w = 0; w < WORDS; w++) if (nm[w]) atomicOr(&NM[(size_t)m * words + w], nm[w]);
        }
        FDC_DT(3);
        __syncthreads();
        FDC_DT(4);
        // ---- the list of the next slab: who is still alive, in the reference's list order (old ones, then by activation)
        const int nt = cnt[5] < kDetMaxCells ? cnt[5] : kDetMaxCells;
        if (tid < nt) {
            const int me = TL[A_KEY + tid];
            int d = 0;
            for (int u = 0; u < nt; u++) d += TL[A_KEY + u] < me ? 1 : 0;
            AL[A_T + d] = TL[A_T + tid]; AL[A_KEY + d] = me; AL[A_DS + d] = TL[A_DS + tid]; AL[A_DE + d] = TL[A_DE + tid]; AL[A_STK + d] = TL[A_STK + tid];
        }
        nal = nt;
        __syncthreads();
        FDC_DT(5);
    }
    // the channels that outlive the call: their misses in a row go into the next call's list
    for (int i = tid; i < nal; i += 64 * kDetWaves) chs[AL[A_T + i]].streak = AL[A_STK + i];
    __syncthreads();

    // ---- sequence numbers of the new channels: block of activation, then candidate order inside the block
    const int nnew = cnt[2], nown = nlive0 + nnew;
    {
        int acc = 0;
        for (int m0 = 0; m0 < nb; m0 += 64 * kDetWaves) {
            const int m = m0 + tid;
            int n = 0;
            if (m < nb) for (int w = 0; w < words; w++) n += __popcll(NM[(size_t)m * words + w]);
            long long tot;
            const int ex = (int)block_exscan(n, &tot, sh);
            if (m < nb) NB[m] = acc + ex;
            acc += (int)tot;
        }
    }
    __syncthreads();

    // ---- the records of every channel (a wave each, any order)
    auto emit = [&](const DetChanRegs &c, const int end, const int streak) {
        const int w_ = 1 << c.cls, len = w_ - w_ / dp.R;
        const int first = c.a < 0 ? 0 : c.a + 1;              // first block in which the channel is matched
        // stream bookkeeping: Q(m) = buffered before the call + blocks extracted up to block m; E = emitted prefix
        // events: a new channel's activation block counts as one (two blocks buffered), every further block as one
        auto qof = [&](int m) { return c.buf0 + (c.a < 0 ? m + 1 : m - c.a + 2); };            // stream length after block m
        auto divmb = [&](int q) { return dp.mb_shift >= 0 ? q >> dp.mb_shift : q / mb; };     // maxblocks >= 2 (a shift when it is 2^k)
        auto eof = [&](int q) { return mb < 0 ? 0 : mb == 0 ? q : mb == 1 ? (q > 0 ? q - 1 : 0) : divmb(q) * mb; };   // emitted after a block that left q
        auto nem = [&](int m) {                                 // partial emissions up to and including block m (this call)
            if (mb < 0) return 0;
            const int ev = c.a < 0 ? m + 1 : m - c.a + 1;     // events so far
            return mb <= 1 ? ev : divmb(qof(m));
        };
        auto take = [&](int n) {                                // n places in the segment's record list
            int p = 0;
            if (lane == 0) p = atomicAdd(&cnt[3], n);
            return __builtin_amdgcn_readfirstlane(p);
        };
        // the activation block itself (no matching there, but the partial check behind process_channel_hist, :317 / :359)
        if (c.a >= 0) {
            const int q = qof(c.a), e1 = eof(q);
            if (mb >= 0 && e1 > 0) {
                const int p = take(1);
                if (lane == 0) {
                    SinkPdu r;
                    r.key = ((long long)c.a << 40) | (1ll << 39) | ((long long)sg << 28) | ((long long)(sd ? 1 : 0) << 27) | ((long long)c.own * 2 + 1);
                    r.act_time = ((long long)c.thi << 32) | (unsigned)c.tlo; r.off = 0;
                    r.owner = ow0 + c.own; r.q0 = 0; r.q1 = e1; r.count = 2; r.chan_id = c.id; r.part = c.part0;
                    r.flags = (c.cls << 8) | (1 << 16); r.vstart = c.es;
                    pl[p] = r;
                }
            }
        }
        // partial emissions of the blocks the channel is extracted from (those before `end`)
        const int lim = end >= 0 ? end : nb;
        if (mb >= 0)
            for (int m0 = first; m0 < lim; m0 += 64) {
                const int m = m0 + lane;
                const int q = qof(m), e1 = eof(q), qp = m == first ? (c.a < 0 ? c.buf0 : qof(c.a)) : qof(m - 1), e0 = m == first && c.a < 0 ? 0 : eof(qp);
                const bool em = m < lim && e1 > e0;
                const unsigned long long EM = __ballot(em);
                if (!EM) continue;
                const int p = take(__popcll(EM));
                if (em) {
                    SinkPdu r;
                    r.key = ((long long)m << 40) | (1ll << 39) | ((long long)sg << 28) | ((long long)(sd ? 1 : 0) << 27) | ((long long)c.own * 2 + 1);
                    r.act_time = ((long long)c.thi << 32) | (unsigned)c.tlo; r.off = 0;
                    r.owner = ow0 + c.own; r.q0 = e0; r.q1 = e1; r.count = c.cnt0 + (c.a < 0 ? m + 1 : m - c.a + 2); r.chan_id = c.id;
                    r.part = c.part0 + nem(m) - 1;
                    r.flags = (c.cls << 8) | (1 << 16); r.vstart = c.es;
                    pl[p + __popcll(EM & ltm)] = r;
                }
            }
        // what the call leaves of the channel
        const int firstproc = c.a < 0 ? 0 : c.a;                // first block it was extracted from, if any
        const int lastm = end >= 0 ? end - 1 : nb - 1;          // last one (before firstproc: none in this call)
        const bool any = lastm >= firstproc;
        const int Qf = any ? qof(lastm) : c.buf0;
        const int Ef = any ? eof(Qf) : 0;                        // partial emissions of this call
        const int npart = any ? nem(lastm) : 0;
        const int cntf = c.cnt0 + (Qf - c.buf0);
        int p = 0;
        if (end >= 0) p = take(1);
        if (lane == 0) {
            SinkOwner o{};
            o.len = len; o.cls = c.cls; o.carried = c.buf0; o.total = Qf; o.emitted = end >= 0 ? Qf : Ef; o.prev_off = c.prev_off;
            o.slot0 = c.a < 0 ? 1 : c.a; o.phase0 = c.phase0; o.pinc = c.es & rm; o.estart = c.es; o.win0 = wofs[c.cls];
            ow[c.own] = o;
            if (end >= 0) {                                      // emit_channel(), :406-452: everything that is left
                SinkPdu r;
                r.key = ((long long)end << 40) | (1ll << 39) | ((long long)sg << 28) | ((long long)c.own * 2);
                r.act_time = ((long long)c.thi << 32) | (unsigned)c.tlo; r.off = 0;
                r.owner = ow0 + c.own; r.q0 = Ef; r.q1 = Qf; r.count = cntf; r.chan_id = c.id; r.part = c.part0 + npart;
                r.flags = 1 | (c.cls << 8) | (1 << 16); r.vstart = c.es;
                pl[p] = r;
            } else {                                             // still alive: an entry of the list the next call starts from (placed below)
                int d = atomicAdd(&cnt[4], 1);
                if (d >= kDetMaxCells) { d = kDetMaxCells - 1; atomicExch(error, 1); }   // live channels are disjoint, at most one per cell — unless
                                                                                          // a threshold below 0 dB lets one-cell candidates through
                SV[d] = c.own;
                L2[DC_ID * kDetMaxCells + d] = c.id; L2[DC_DSTART * kDetMaxCells + d] = c.ds; L2[DC_DSTOP * kDetMaxCells + d] = c.de;
                L2[DC_ESTART * kDetMaxCells + d] = c.es; L2[DC_CLS * kDetMaxCells + d] = c.cls; L2[DC_COUNT * kDetMaxCells + d] = cntf;
                L2[DC_PHASE * kDetMaxCells + d] = (cntf * (c.es & rm)) & rm; L2[DC_PINC * kDetMaxCells + d] = c.es & rm;
                L2[DC_INACT * kDetMaxCells + d] = streak; L2[DC_PART * kDetMaxCells + d] = c.part0 + npart;
                L2[DC_OWNER * kDetMaxCells + d] = c.own; L2[DC_TAIL * kDetMaxCells + d] = 0;
                L2[DC_TIME_LO * kDetMaxCells + d] = c.tlo; L2[DC_TIME_HI * kDetMaxCells + d] = c.thi;
            }
        }
    };
    FDC_DT(6);
    for (int t = wv; t < nown; t += kDetWaves) {
        const DetCh h = chs[t];
        DetChanRegs c;
        if (t < nlive0) {
            c.id = Lg[DC_ID * kDetMaxCells + t]; c.ds = Lg[DC_DSTART * kDetMaxCells + t]; c.de = Lg[DC_DSTOP * kDetMaxCells + t];
            c.es = Lg[DC_ESTART * kDetMaxCells + t]; c.cls = Lg[DC_CLS * kDetMaxCells + t]; c.cnt0 = Lg[DC_COUNT * kDetMaxCells + t];
            c.inact0 = Lg[DC_INACT * kDetMaxCells + t]; c.part0 = Lg[DC_PART * kDetMaxCells + t]; c.buf0 = Lg[DC_TAIL * kDetMaxCells + t];
            c.phase0 = Lg[DC_PHASE * kDetMaxCells + t]; c.tlo = Lg[DC_TIME_LO * kDetMaxCells + t]; c.thi = Lg[DC_TIME_HI * kDetMaxCells + t];
            c.prev_off = Og[t]; c.a = -1; c.own = t;
        } else {
            int rk = NB[h.a];
            for (int w = 0; w < (h.j >> 6); w++) rk += __popcll(NM[(size_t)h.a * words + w]);
            rk += __popcll(NM[(size_t)h.a * words + (h.j >> 6)] & ((1ull << (h.j & 63)) - 1ull));
            c.id = counter0 + rk; c.ds = h.ds; c.de = h.de; c.es = h.es; c.cls = h.cls; c.cnt0 = 0; c.inact0 = 0; c.part0 = 0; c.buf0 = 0;
            c.phase0 = 0; c.tlo = (int)(unsigned)(now & 0xFFFFFFFFll); c.thi = (int)(now >> 32); c.prev_off = 0; c.a = h.a; c.own = nlive0 + rk;
        }
        emit(c, h.end, h.streak);
    }
    __syncthreads();
    // ---- the list of the next call: the survivors in sequence order (old ones first, in their old order, then by activation)
    const int nsv = cnt[4] < kDetMaxCells ? cnt[4] : kDetMaxCells;
    if (tid < nsv) {
        const int me = SV[tid];
        int d = 0;
        for (int u = 0; u < nsv; u++) d += SV[u] < me ? 1 : 0;
        SVS[tid] = d;
    }
    __syncthreads();
    for (int e = tid; e < nsv * kDetFields; e += 64 * kDetWaves) {
        const int f = e / nsv, u = e - f * nsv;
        Lg[f * kDetMaxCells + SVS[u]] = L2[f * kDetMaxCells + u];
    }
    if (tid == 0) {
        sst[sg].nlive = nsv; sst[sg].counter = counter0 + nnew;
        npdu[lst] = cnt[3]; nowner[sg] = nown;
#ifdef FDC_DET_STATS
        printf("[det]   wave 0 of seg %d: %d regions set up in %llu kcycles, %d passes of alive channels in %llu, %d activations in %llu\n", sg, wcnt[0],
               wacc[0] / 1000, wcnt[1], wacc[1] / 1000, wcnt[2], wacc[2] / 1000);
        FDC_DT(7);
        printf("[det] seg %d: %d cells, %d live before, %d new, %d survive, %d records, %d regions in all slabs; kcycles wave 0: tables %llu cover %llu regions %llu work %llu wait %llu "
               "rebuild %llu prefix %llu emit+rest %llu\n", sg, g.ncell, nlive0, nnew, nsv, cnt[3], nregsum, tacc[0] / 1000, tacc[1] / 1000, tacc[2] / 1000,
               tacc[3] / 1000, tacc[4] / 1000, tacc[5] / 1000, tacc[6] / 1000, tacc[7] / 1000);
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
// do the tracker's tables (per block of a call: activation masks and their prefix) fit the LDS?  >= 0: yes
int det_track_staged(int nbmax, int max_cand_cap)
{
    const int w0 = (max_cand_cap + 63) / 64, words = w0 <= 2 ? w0 : 8;      // the kernel exists for 1, 2 and 8 words
    if (w0 > 8) return -1;
    return det_lds_bytes(nbmax, words) <= (size_t)150 * 1024 ? kDetStaged : -1;   // does not fit: the bank takes the host engine
}
hipError_t launch_det_track(const DetParams &dp, int nb, const DetGeom *geom, DetSegState *sst, int32_t *live, int64_t *live_off,
                            const int2 *cand, const int64_t *cand_base, const int32_t *ncand, const int32_t *win_off, long long now,
                            SinkPdu *pdus, const int64_t *pdu_base, int32_t *npdu, SinkOwner *owners, const int64_t *owner_base,
                            int32_t *nowner, DetCh *chs, int32_t *live2, int32_t *error, hipStream_t s)
{
    if (dp.nseg <= 0) return hipSuccess;
    const int words = (dp.max_cand_cap + 63) / 64;
    if (det_track_staged(dp.nbmax, dp.max_cand_cap) < 0 || words > 8) return hipErrorInvalidValue;
    DetParams dq = dp;
    // n / dec by a multiply-high: exact while n * dec < 2^32 (n <= N); otherwise the kernel divides
    dq.dec_magic = (dp.dec > 1 && (unsigned long long)dp.N * (unsigned long long)dp.dec < (1ull << 32)) ? (unsigned)((1ull << 32) / (unsigned)dp.dec) + 1u : 0u;
#define FDC_LT(W) \
    hipLaunchKernelGGL(k_det_track<W>, dim3((unsigned)dp.nseg), dim3(64 * kDetWaves), det_lds_bytes(nb, W), s, dq, nb, geom, sst, live, live_off, \
                       cand, cand_base, ncand, win_off, now, pdus, pdu_base, npdu, owners, owner_base, nowner, chs, live2, error)
    if (words == 1) FDC_LT(1); else if (words == 2) FDC_LT(2); else FDC_LT(8);
#undef FDC_LT
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
    __shared__ int ncar, maxlist;
    const int tid = threadIdx.x;
    if (tid < 32) hist[tid] = 0;
    if (tid == 0) { ncar = 0; maxlist = 0; }
    __syncthreads();
    long long accA = 0, accB = 0;
    int total_owner = 0, max_region = 0;
    for (int rg = 0; rg <= nseg; rg++) {
        const long long o0 = rg == 0 ? 0 : owner_base[rg - 1];
        const int cnt = rg == 0 ? npac : nowner[rg - 1];
        total_owner += cnt;
        max_region = cnt > max_region ? cnt : max_region;
        for (int c0 = 0; c0 < cnt; c0 += 1024) {
            const int c = c0 + tid;
            long long a = 0, b = 0;
            if (c < cnt) {
                const SinkOwner o = owners[o0 + c];
                a = (long long)o.emitted * o.len; b = (long long)(o.total - o.emitted) * o.len;
                if (o.total > o.carried) atomicAdd(&hist[o.cls & 31], o.total - o.carried);     // extractions of the call, per width class
                if (o.carried > 0) atomicAdd(&ncar, 1);
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
        if (l < nlist && ntask[l] > 0) atomicMax(&maxlist, ntask[l]);
        __syncthreads();
        // a thread per record (a bank of 256 channels has 256 short lists: a wave per list would walk them sixteen deep, two dependent
        // loads each); the list of a record by bisection of the bases
        const int lim = nlist - l0 < 1024 ? nlist - l0 : 1024;
        for (int idx = tid; idx < (int)tot; idx += 1024) {
            const int g = np + idx;
            int lo = 0, hi = lim - 1;                           // last list whose base is <= g
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (lbase[mid] <= g) lo = mid; else hi = mid - 1; }
            SinkPdu r = pdus[pdu_base[l0 + lo] + (g - lbase[lo])];
            const SinkOwner o = owners[r.owner];
            r.off = o.a_off + (long long)r.q0 * o.len;
            pdus_out[g] = r;
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
        sum->max_list_tasks = maxlist; sum->ncarry = ncar; sum->max_region_owners = max_region; sum->pad = 0;
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

// The layout summary and the first emission records go to the host by STORES of a kernel into pinned host memory, not by copies:
// a copy would queue on the DMA engine behind the previous batch's payload (tens of MB, 1.6 ms and more) and hold back the
// host's one wait of the call — and with it this batch's extractions — until that payload has left (measured: the step was
// payload copy + extraction kernels, strictly one after the other).
__global__ __launch_bounds__(256) void k_sink_publish(const SinkSummary *__restrict__ sum, const SinkPdu *__restrict__ pdus,
                                                      SinkSummary *__restrict__ h_sum, SinkPdu *__restrict__ h_pdus, int eager)
{
    static_assert(sizeof(SinkSummary) % 4 == 0 && sizeof(SinkPdu) % 8 == 0, "word copies");
    const int tid = blockIdx.x * 256 + threadIdx.x, nthr = gridDim.x * 256;
    const int n = sum->npdu < eager ? sum->npdu : eager;
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(pdus);
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(h_pdus);
    for (long long i = tid; i < (long long)n * (long long)(sizeof(SinkPdu) / 8); i += nthr) dst[i] = src[i];
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < (int)(sizeof(SinkSummary) / 4); i += 256)
            reinterpret_cast<int32_t *>(h_sum)[i] = reinterpret_cast<const int32_t *>(sum)[i];
}

hipError_t launch_sink_publish(const SinkSummary *sum, const SinkPdu *pdus, SinkSummary *h_sum, SinkPdu *h_pdus, int eager, hipStream_t s)
{
    hipLaunchKernelGGL(k_sink_publish, dim3(8), dim3(256), 0, s, sum, pdus, h_sum, h_pdus, eager);
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
    // Places: one atomic per WORKGROUP and width class on the class counters (a bank of equal channels has one class: a counter per task
    // serialised at 11 ns each, one per wave still at 3 000 atomics on one address), the waves of a workgroup share through LDS.
    __shared__ int wcnt[32], wbase[32];
    if (threadIdx.x < 32) wcnt[threadIdx.x] = 0;
    __syncthreads();
    int woff = 0, rank = 0;
    unsigned long long todo = __ballot(have);
    const unsigned long long lt = lanemask_lt();
    while (todo) {
        const int c0 = __builtin_amdgcn_readlane(t.cls, __builtin_ctzll(todo));
        const unsigned long long peers = __ballot(have && t.cls == c0) & todo;
        int base = 0;
        if ((threadIdx.x & 63) == __builtin_ctzll(peers)) base = atomicAdd(&wcnt[c0 & 31], __popcll(peers));
        base = __builtin_amdgcn_readlane(base, __builtin_ctzll(peers));
        if (have && t.cls == c0) { woff = base; rank = __popcll(peers & lt); }
        todo &= ~peers;
    }
    __syncthreads();
    if (threadIdx.x < 32 && wcnt[threadIdx.x] > 0) wbase[threadIdx.x] = atomicAdd(&class_fill[threadIdx.x], wcnt[threadIdx.x]);
    __syncthreads();
    const int pos = have ? sum->class_base[t.cls & 31] + wbase[t.cls & 31] + woff + rank : 0;
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
constexpr int kCarrySplit = 64;     // workgroups per stream (a wide carrier buffers megabytes)
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
    // a few streams carry everything (a wide channel: megabytes): many workgroups per stream, 16 bytes per lane and trip
    if (!((o.len | o.prev_off | o.a_off | o.b_off) & 1)) {
        const float4 *s4 = reinterpret_cast<const float4 *>(src);
        float4 *a4 = reinterpret_cast<float4 *>(da), *b4 = reinterpret_cast<float4 *>(db);
        const long long n4 = n >> 1, nA4 = nA >> 1;
        for (long long i = (long long)blockIdx.z * 256 + threadIdx.x; i < n4; i += 256 * kCarrySplit) (i < nA4 ? a4 : b4)[i] = s4[i];
    } else {
        for (long long i = (long long)blockIdx.z * 256 + threadIdx.x; i < n; i += 256 * kCarrySplit) (i < nA ? da : db)[i] = src[i];
    }
}

// nowner_max: the largest number of streams one region of the owner table holds in this call (the grid's width)
hipError_t launch_carry_copy(const SinkOwner *owners, int nowner_max, const int64_t *owner_base, const int32_t *nowner, int npac, int nseg,
                             const SinkSummary *sum, const float2 *prev, float2 *cur, hipStream_t s)
{
    (void)sum;
    if (nowner_max <= 0 || !prev) return hipSuccess;
    hipLaunchKernelGGL(k_carry_copy, dim3((unsigned)nowner_max, (unsigned)(nseg + 1), kCarrySplit), dim3(256), 0, s, owners, owner_base,
                       nowner, npac, prev, cur);
    return hipGetLastError();
}

hipError_t init_sink_kernels()
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_det_track<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_det_track<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_det_track<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
    return e;
}

}  // namespace fdc
