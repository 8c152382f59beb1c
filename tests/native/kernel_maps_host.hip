// The index arithmetic of the kernels, compiled for the HOST from the very headers the kernels include (hipcc, no GPU needed):
// nvmolkit_amd/csrc/tile_maps.h (workgroup -> tile maps of the matrix-core kernels) and nvmolkit_amd/csrc/hess_pass.h (row
// offsets of the packed inverse Hessian, LDS layout and residency rule of the BFGS kernels, in their four workgroup sizes).
// tests/test_tile_maps.py and tests/test_hess_layout.py drive these entry points.
#include "../../nvmolkit_amd/csrc/tile_maps.h"

#define NVMK_BFGS_NS t512
#define NVMK_BFGS_THREADS 512
#include "../../nvmolkit_amd/csrc/hess_pass.h"
#undef NVMK_BFGS_NS
#undef NVMK_BFGS_THREADS
#define NVMK_BFGS_NS t256
#define NVMK_BFGS_THREADS 256
#include "../../nvmolkit_amd/csrc/hess_pass.h"
#undef NVMK_BFGS_NS
#undef NVMK_BFGS_THREADS
#define NVMK_BFGS_NS t128
#define NVMK_BFGS_THREADS 128
#include "../../nvmolkit_amd/csrc/hess_pass.h"
#undef NVMK_BFGS_NS
#undef NVMK_BFGS_THREADS
#define NVMK_BFGS_NS t64
#define NVMK_BFGS_THREADS 64
#include "../../nvmolkit_amd/csrc/hess_pass.h"
#undef NVMK_BFGS_NS
#undef NVMK_BFGS_THREADS

#include <cstdint>

using namespace nvmk::minim;

extern "C" {

unsigned chk_dense_super_m(long long tilesM) { return nvmk::maps::dense_super_m(tilesM); }
void chk_dense_tile(unsigned bx, unsigned by, unsigned tilesN, unsigned superM, unsigned* tm, unsigned* tn) {
  nvmk::maps::dense_tile(bx, by, tilesN, superM, *tm, *tn);
}
int chk_symmetric_supertile(unsigned long long sidx, unsigned superN, unsigned* sm, unsigned* sn) {
  return nvmk::maps::symmetric_supertile(sidx, superN, *sm, *sn) ? 1 : 0;
}
void chk_count_tile(unsigned bx, unsigned sm, unsigned sn, unsigned superH, unsigned superW, unsigned* tm, unsigned* tn) {
  nvmk::maps::count_tile(bx, sm, sn, superH, superW, *tm, *tn);
}

int chk_panel_of(unsigned block, unsigned round, unsigned slots, unsigned lo, unsigned hi, unsigned* pnl) {
  return nvmk::maps::panel_of(block, round, slots, lo, hi, *pnl) ? 1 : 0;
}
int chk_panel_round_exists(unsigned round, unsigned slots, unsigned lo, unsigned hi) { return nvmk::maps::panel_round_exists(round, slots, lo, hi) ? 1 : 0; }
int chk_panel_pending_after(int chunksPerTile, int dist, int ch) { return nvmk::maps::panel_pending_after(chunksPerTile, dist, ch); }

int64_t chk_hess_row_offset(int64_t r) { return t256::hess_row_offset(r); }
int     chk_hess_row_offset32(int r) { return t64::hess_row_offset32(r); }
int64_t chk_lds_vector_doubles(int threads, int64_t n) {
  return threads == 64 ? t64::lds_vector_doubles(n) : threads == 128 ? t128::lds_vector_doubles(n) : threads == 512 ? t512::lds_vector_doubles(n) : t256::lds_vector_doubles(n);
}
int64_t chk_lds_hessian_doubles(int threads, int64_t ldsDoubles, int64_t n) {
  return threads == 64 ? t64::lds_hessian_doubles(ldsDoubles, n)
         : threads == 128 ? t128::lds_hessian_doubles(ldsDoubles, n)
         : threads == 512 ? t512::lds_hessian_doubles(ldsDoubles, n)
                          : t256::lds_hessian_doubles(ldsDoubles, n);
}
int chk_resident_rows(int threads, int n, int64_t hldsDoubles) {
  return threads == 64 ? t64::resident_rows(n, hldsDoubles) : threads == 128 ? t128::resident_rows(n, hldsDoubles) : threads == 512 ? t512::resident_rows(n, hldsDoubles) : t256::resident_rows(n, hldsDoubles);
}
int64_t chk_tail_pad_doubles() { return t256::kHessTailPadDoubles; }

}  // extern "C"
