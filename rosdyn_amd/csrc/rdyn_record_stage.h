// rdyn_record_stage.h -- the SAMPLE-MAJOR ("drop-in") records of the kinematic / torque / inertia kernels written in whole lines.
//
// In RDYN_LAYOUT_SAMPLE_MAJOR a sample's record (VectorOfAffine3d / VectorOfVector6d / Matrix6Xd / MatrixXd / VectorXd images:
// primitives_impl.h:884-912, 927-949, 981-1013, 1264-1293, 1357-1379) is contiguous and a wave's 64 records are ONE run of 64 rb bytes,
// but the lane that computes a value owns one 8-byte piece of it at a stride of rb bytes: stored from that lane a store instruction
// touches 64 lines with 8 bytes each (round 5: every kinematic kernel did).  Here every value goes through WAVE-PRIVATE LDS and leaves
// 16 bytes per lane with the lanes running along the bytes of a record, whole 128-byte lines only, nontemporal (written once, never
// re-read) -- the copy-out of the regressor image kernel (rdyn_image_impl.h) for records that are produced
//   * link by link (frames, twists, acceleration / jerk twists: PIECE bytes per link)  -> RecordRing<PIECE>
//   * at once at the end of the sweep (tool frame, Jacobian, joint torque, inertia)     -> SmallRecords
// Only wave-local ordering is needed (no barrier).  The INPUTS keep their per-lane loads: reading a wave's 64 input records as whole
// lines into an LDS tile was built and measured (profiles/r6/sweep_sheet_input_tile.txt) -- every kernel the same or slower (getTwist
// 91 -> 101 us, getWrench with its external wrenches through the record tile 234 -> 262 us): these kernels move 0.3-0.8 KB per sample
// and are bound by the LIFETIME of a wave, not by the address unit; one more LDS hop in front of the first sincos costs more than the
// strided loads do (and a load loop that waits per trip costs a whole memory round trip per trip: +27 us on a 129 us kernel).
// PERSISTENT waves (a few workgroups per CU walking the batch in strides of the grid, the next tile's inputs requested while the current
// one is swept; profiles/r6/sweep_sheet_persistent.txt) were built and measured too: slower everywhere (getTwist 91 -> 129 us,
// getTransformations 129 -> 166 us) -- the hardware's own dispatch keeps more waves in flight than a fixed grid does.
// Preconditions, checked by the host (rdyn_api.cpp: natural strides, base pointer
// 128-byte aligned) and by the kernel (a full wave): everything else keeps the 8-byte stores.
#ifndef RDYN_RECORD_STAGE_H
#define RDYN_RECORD_STAGE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace
{
typedef double rs_d2u __attribute__((ext_vector_type(2), aligned(8)));
typedef double rs_d2a __attribute__((ext_vector_type(2), aligned(16)));

__device__ __forceinline__ void rs_wave_fence()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Records that grow by PIECE bytes per link (PIECE = 96: a 3x4 frame, 48: a 6-vector).  Per sample one LDS slot:
//   [0, 112)        the record's first 112 bytes, kept to the end: the head of a record (at most 112 bytes: misalignments are multiples
//                   of 16) shares its line with the tail of the previous sample's record (records are rb bytes apart, not a multiple of
//                   128 in general)
//   [112, 112 + W)  a ring over the rest, addressed by record offset: what a flush has not written yet (< 128 bytes that wait for
//                   their line to complete) plus the next piece
// After every link the lines of every record that are complete by now are written; finish() writes the lines two records share.
// Sample i of the wave starts at byte i rb of the run; its misalignment m_i = (i rb) mod 128 depends on i mod 8 only (rb is a multiple
// of 16): the copy-out role of a lane -- 8 lanes per line, 8 samples per store instruction, 8 instructions per flush -- keeps ONE
// alignment class, so the piece arithmetic runs once per flush.
template <int PIECE>
struct RecordRing
{
  static_assert(PIECE % 16 == 0 && PIECE >= 16 && PIECE <= 128, "a flush completes at most one line per record");
  static constexpr int W = PIECE + 112;
  static constexpr int HEAD = 112;  // a record's head is at most 112 bytes (misalignments are multiples of 16)
  static constexpr int SLOT = HEAD + W;
  // odd number of 16-byte units: the 8-byte staging writes of consecutive lanes fall into different LDS banks
  static constexpr int PITCH = SLOT + (((SLOT / 16) & 1) ? 0 : 16);
  static constexpr int BYTES = 64 * PITCH;  // LDS per wave

  char* area;    // this wave's staging area (BYTES)
  char* mine;    // the lane's own slot
  char* ywave;   // the wave's 64 records in memory (line aligned)
  uint32_t rb;   // bytes per record
  int sub, cj;   // copy-out role: chunk cj of the line of sample 8 it + sub

  __device__ __forceinline__ void init(char* lds_area, void* wave_records, uint32_t record_bytes, int lane)
  {
    area = lds_area;
    mine = lds_area + lane * PITCH;
    ywave = (char*)wave_records;
    rb = record_bytes;
    sub = lane >> 3;
    cj = lane & 7;
  }
  __device__ static __forceinline__ uint32_t pos(uint32_t x) { return x < (uint32_t)HEAD ? x : (uint32_t)HEAD + (x - (uint32_t)HEAD) % (uint32_t)W; }
  // value at record byte x (a multiple of 8) of the lane's own sample
  __device__ __forceinline__ void put(uint32_t x, double v) const { *(double*)(mine + pos(x)) = v; }
  // every record's bytes [Ep, E) are staged (E - Ep <= PIECE): write the lines this completes
  __device__ __forceinline__ void flush(int Ep, int E) const
  {
    rs_wave_fence();
    const int m = (int)(((uint32_t)sub * rb) & 127u);
    int lo = Ep - ((m + Ep) & 127);
    if (lo < 0) lo += 128;  // (no line boundary of this record at or below Ep: the first one is the end of its head)
    const int hi = E - ((m + E) & 127);
    if (hi > lo)
    {
      const uint32_t y = (uint32_t)(lo + 16 * cj);
      const char* src = area + sub * PITCH + pos(y);
      char* dst = ywave + (uint32_t)sub * rb + y;
      const uint32_t dstep = 8u * rb;
#pragma unroll
      for (int it = 0; it < 8; ++it)
      {
        const rs_d2a v = *(const rs_d2a*)(src + it * (8 * PITCH));
        __builtin_nontemporal_store((rs_d2u)v, (rs_d2u*)(dst + (uint32_t)it * dstep));
      }
    }
    rs_wave_fence();
  }
  // all rb bytes of every record are staged and flushed: the lines shared by records i and i + 1 (tail of i, head of i + 1)
  __device__ __forceinline__ void finish() const
  {
    rs_wave_fence();
    const int mn = (int)(((uint32_t)(sub + 1) * rb) & 127u);  // bytes of the shared line that belong to record 8 it + sub (0: none; sub = 7: always 0)
    if (mn)
    {
      const int b = 16 * cj;
      const bool tail = b < mn;
      const uint32_t y = tail ? rb - (uint32_t)mn + (uint32_t)b : (uint32_t)(b - mn);
      const char* src = area + (sub + (tail ? 0 : 1)) * PITCH + pos(y);
      char* dst = ywave + (uint32_t)(sub + 1) * rb - (uint32_t)mn + (uint32_t)b;
      const uint32_t dstep = 8u * rb;
#pragma unroll
      for (int it = 0; it < 8; ++it)
      {
        const rs_d2a v = *(const rs_d2a*)(src + it * (8 * PITCH));
        __builtin_nontemporal_store((rs_d2u)v, (rs_d2u*)(dst + (uint32_t)it * dstep));
      }
    }
    rs_wave_fence();
  }
};

// Records of `rec` doubles that a lane holds complete at the end of its sweep: every lane drops its record into the wave's tile
// (pitch rec | 1 doubles: consecutive lanes on different banks), then the wave copies the run of 64 rec doubles out, 16 bytes per lane,
// whole lines.  LDS: 64 (rec | 1) doubles.
struct SmallRecords
{
  double* tile;
  double* mine;
  int rec, prec;
  __device__ __forceinline__ void init(char* lds_area, int rec_doubles, int lane)
  {
    tile = (double*)lds_area;
    rec = rec_doubles;
    prec = rec_doubles | 1;
    mine = tile + lane * prec;
  }
  __device__ __forceinline__ void put(int e, double v) const { mine[e] = v; }
  // ns = 64: the tile holds the wave's 64 records; ns = 32: a HALF-wave's (the tile is half the size -- twice the waves per CU where the
  // LDS bounds them -- and the wave stages and copies out its two halves one after the other: 32 rec doubles are whole lines too)
  __device__ __forceinline__ void copy_out(void* records, int lane, int ns = 64) const
  {
    rs_wave_fence();
    // chunk c = it * 64 + lane holds doubles 2 c, 2 c + 1 of the run; double d belongs to sample d / rec, element d % rec
    int s0 = (int)(((float)(2 * lane) + 0.5f) / (float)rec);
    int e0 = 2 * lane - s0 * rec;
    const int ds = 128 / rec, de = 128 - ds * rec;  // wave-uniform
    char* dst = (char*)records + lane * 16;
#pragma unroll 4
    for (int it = 0; it < (ns * rec + 127) / 128; ++it)
    {
      if (2 * (it * 64 + lane) < ns * rec)
      {
        int s1 = s0, e1 = e0 + 1;
        if (e1 == rec)
        {
          e1 = 0;
          ++s1;
        }
        rs_d2u v;
        v.x = tile[s0 * prec + e0];
        v.y = tile[s1 * prec + e1];
        __builtin_nontemporal_store(v, (rs_d2u*)dst);
      }
      dst += 1024;
      s0 += ds;
      e0 += de;
      if (e0 >= rec)
      {
        e0 -= rec;
        ++s0;
      }
    }
    rs_wave_fence();
  }
};

// The INPUT side for records that are not on the wave's critical path (the external wrenches of getWrench: 6 (NJ + 1) doubles per sample,
// needed link by link behind each link's kinematics): the wave reads its 64 records as ONE run, 16 bytes per lane, all loads in flight
// before the first LDS write, into the same [sample][rec | 1] tile the wrenches are parked in -- 21 whole-line loads instead of 42
// loads that touch 64 lines each: getWrench 245 -> 204 us per 1e6.  (The joint inputs stay per-lane loads: q feeds the first sincos, and
// Dq / DDq through a tile -- made visible behind the first joint's sincos -- cost the torque kernel 83 -> 88 us: measured, not shipped.)
template <int MAX_IT>
__device__ __forceinline__ void load_records_into_tile(double* tile, int prec, const double* wave_run, int rec, int lane)
{
  const int s_first = (int)(((float)(2 * lane) + 0.5f) / (float)rec);
  const int e_first = 2 * lane - s_first * rec;
  const int ds = 128 / rec, de = 128 - ds * rec;  // wave-uniform
  rs_d2a v[MAX_IT];
#pragma unroll
  for (int it = 0; it < MAX_IT; ++it)
    if (2 * (it * 64 + lane) < 64 * rec) v[it] = __builtin_nontemporal_load((const rs_d2a*)((const char*)wave_run + lane * 16 + it * 1024));
  int s0 = s_first, e0 = e_first;
#pragma unroll
  for (int it = 0; it < MAX_IT; ++it)
  {
    if (2 * (it * 64 + lane) < 64 * rec)
    {
      int s1 = s0, e1 = e0 + 1;
      if (e1 == rec)
      {
        e1 = 0;
        ++s1;
      }
      tile[s0 * prec + e0] = v[it].x;
      tile[s1 * prec + e1] = v[it].y;
    }
    s0 += ds;
    e0 += de;
    if (e0 >= rec)
    {
      e0 -= rec;
      ++s0;
    }
  }
}

}  // namespace
#endif
