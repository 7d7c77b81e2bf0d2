// temporary: stages B/C not built yet
#include "gp_internal.h"
namespace gp {
static int ni() { set_error("stage not implemented yet"); return GP_ERR_STATE; }
int stage_bc_prepare(gp_job*) { return GP_OK; }
void stage_bc_release(gp_job*) {}
int stage_b_reset(gp_job*) { return ni(); }
int stage_b_pass(gp_job*, uint32_t) { return ni(); }
int stage_b_guard(gp_job*) { return ni(); }
int stage_mask(gp_job*) { return ni(); }
int stage_c_run(gp_job*) { return ni(); }
int stage_upper(gp_job*) { return ni(); }
int stage_lengths(gp_job*, uint64_t*, uint8_t*, uint64_t*) { return ni(); }
int stage_sequences(gp_job*, uint8_t*, uint64_t) { return ni(); }
}
