// sppm_kernels.hip -- the translation unit of libsppm_hip.so: the gfx950 kernels (included below, one header per
// stage) and the host side / C ABI (include/sppm_hip.h) of the SPPM path.
//
// Data layout in HBM (P = pixels = hitpoint slots, R = photon-hit records of one batch):
//   hitpoints, SoA by component, indexed by image index:
//     hp_pos[3][P] hp_nrm[3][P] hp_w[3][P] hp_flux[3][P] ld[3][P]  (f64)
//     hp_r2[P] hp_dist[P] (f64)  hp_n[P] hp_obj[P] hp_valid[P] (i32)
//     rule 1 only: hp_phi[3][P] hp_tau[3][P] hp_nd[P] (f64) hp_madd[P] (i32)
//   photon-hit records, SoA: key[R] (u64: photon<<20 | depth<<1 | orient), f[9][Rcap] (position,
//     geometric normal, flux).  Traced into the *unsorted* arrays by atomic append, then
//     counting-sorted by hash-grid bucket into the *sorted* arrays each batch.
//
// Why records are gridded and hitpoints gather (instead of hitpoints gridded and photons
// scattering with atomics): the reference's update (src/sppm2.h:391-421) is applied per photon,
// in photon order, and each accepted photon shrinks the hitpoint's radius, which decides whether
// later photons are accepted.  A hitpoint's final state is a sequential function of the
// time-ordered photons near it -- but photon PATHS do not depend on hitpoints.  So photons are
// traced in parallel, and each hitpoint replays "its" photons in key (= photon, bounce) order in
// registers: exact reference semantics, no atomics on hitpoint data, deterministic.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <chrono>
#include <type_traits>
#include <vector>

#include "sppm_device.hpp"

// ================================ kernels ==========================================================

#include "sppm_eye.hpp"
#include "sppm_trace.hpp"
#include "sppm_records.hpp"
#include "sppm_gather_simple.hpp"
#include "sppm_cap.hpp"
#include "sppm_image.hpp"
#include "sppm_exchange.hpp"
#include "sppm_gather_tiles.hpp"
#include "sppm_deposit.hpp"
#include "sppm_deposit_wg.hpp"

// ================================ host side / C ABI =================================================

#include "sppm_host_ctx.hpp"
#include "sppm_host_create.hpp"
#include "sppm_host_eye.hpp"
#include "sppm_host_trace.hpp"
#include "sppm_host_gather.hpp"
#include "sppm_host_cap.hpp"
#include "sppm_host_pass.hpp"
#include "sppm_host_io.hpp"
#include "sppm_host_static.hpp"
