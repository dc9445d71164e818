// The plan object behind asp_sa_plan_create and the host helpers the annealing translation
// units share (csrc/sa_sweep.hip defines them; csrc/sa_shuffled.hip uses them).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "asp_common.hpp"
#include "sa_plan.hpp"

struct asp_sa_plan {
  asp::SaHostLayout host;
  hipStream_t stream = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  float last_sweep_ms = 0.0f, last_total_ms = 0.0f;
  int force_m = 0, force_threads = 0;
  int force_packed = 0;  // asp_sa_set_packed: 0 auto, 1 bits in LDS, 2 bits in HBM
  bool allow_wide = true;  // asp_sa_set_wide
  int last_m = 0, last_threads = 0, last_groups = 0;
  std::vector<int64_t> last_tracked;
  std::vector<uint64_t> last_accepted;
  int num_cus = 256;
  size_t max_lds = 160 * 1024;
  asp::DeviceBuffer<uint32_t> color_block_start, block_width, ell_col, spin_of_pos, pos_of_spin;
  asp::DeviceBuffer<uint32_t> ell_col4;  // columns as LDS byte addresses of the wide layout (if it fits)
  int last_layout = 0;
  asp::DeviceBuffer<uint64_t> ell_off;
  asp::DeviceBuffer<double> ell_val, field_pos;
  // per-call work buffers, grown on demand and kept (a plan is used by one thread at a time)
  asp::DeviceBuffer<double> w_betas, w_partial, w_e;
  asp::DeviceBuffer<uint64_t> w_best, w_x0, w_x0_perm, w_x;
  asp::DeviceBuffer<long long> w_tracked;
  asp::DeviceBuffer<unsigned long long> w_accepted;
  asp::DeviceBuffer<double> w_field_cache;  // [groups][blocks][M][64], see SweepArgs::field_cache
  asp::DeviceBuffer<uint64_t> w_spins;      // [groups][blocks] sign words of the HBM-resident layout
  asp::DeviceBuffer<long long> w_trace;     // [groups * M][sweeps + 1] tracked energies (asp_sa_anneal_trace)
  // team sweep exchange area, FINE-GRAINED device memory (coherent across XCDs without cache
  // maintenance): arrivals u64[teams] | sums i64[teams][6] | abort u32 (+pad) | flips u64[teams][blocks]
  void *team_area = nullptr;
  size_t team_area_bytes = 0;
  ~asp_sa_plan() {
    if (team_area) (void)hipFree(team_area);
  }
  int team_mode = -1;  // asp_sa_set_team: -1 auto, 0 off, G >= 2 forced
  bool use_field_cache = true;
  uint32_t team_abort_host = 0;  // landing place of the watchdog flag's asynchronous read-back
  uint32_t team_watchdog_trips = 0;  // calls of this plan that the team barrier's watchdog cut short (each ~5 s lost)
  // Shuffled sweep (csrc/sa_shuffled.hip; uploaded on first use): rows of A over ORIGINAL
  // indices, padded to whole quads (padding: own index, +0.0) — row i is quads
  // rq_ptr[i] .. rq_ptr[i + 1]; per quad four columns and four values in the interleaving the
  // sweep kernels read (sa_plan.cpp) —, the field in original order and a degree class per spin
  asp::DeviceBuffer<uint32_t> rq_ptr;
  asp::DeviceBuffer<uint32_t> rq_col;   // [quads][4]
  asp::DeviceBuffer<double> rq_val;     // [quads][4]
  asp::DeviceBuffer<double> field_dev;  // [K]
  uint32_t rq_quads = 0, rq_max_quads = 0;
  int shuffled_m = 0, shuffled_waves = 0;  // asp_sa_set_shuffled_launch (0 = automatic)
  int shuffled_teams = 0;                  // asp_sa_set_shuffled_teams (0 = automatic)
  int last_shuffled_levels = 0;            // largest number of levels of the last shuffled call
  uint32_t last_shuffled_log_s = 6, last_shuffled_wgs = 0;  // block size and workgroups of the last shuffled call
  uint32_t last_shuffled_blocks = 0, last_shuffled_quads = 0;  // most blocks / quads of one sweep of that call
  float last_order_ms = 0.0f;              // device time of the last call's order kernels
};


namespace asp {

template <typename T>
int upload_vector(DeviceBuffer<T> &dst, const std::vector<T> &src, hipStream_t stream) {
  ASP_TRY(dst.alloc(src.size()));
  return dst.upload(src.data(), src.size(), stream);
}

// perm[c][b] = sign words (bit = 1: s = -1) of configuration c in the plan's block order, from
// packed original-order configurations x[c][ceil(K/64)] (bit = 1: s = +1); on the plan's stream.
int sa_permute_bits(asp_sa_plan *p, const uint64_t *x, uint32_t count, uint64_t *perm);

// Reported energies of `count` configurations given as block-order sign words;
// partial: count * num_blocks doubles of scratch; on the plan's stream.
int sa_energies_of_perm(asp_sa_plan *p, const uint64_t *perm, uint32_t count, double *partial,
                        double *out_e);

// asp_sa_anneal_batch's items with ASP_SA_BATCH_SHUFFLED set (csrc/sa_shuffled.hip): items[which[k]],
// k < count; adds the device time of their sweeps to *sweep_ms.
int sa_shuffled_batch(asp_sa_batch_item const *items, const uint32_t *which, uint32_t count, float *sweep_ms);

}  // namespace asp
