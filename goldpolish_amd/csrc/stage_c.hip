// stage_c.hip -- placeholder until the Sealer BFS kernels land (next commit).
#include "gp_bc.h"
namespace gp {
int stage_c_prepare(gp_job*) { return GP_OK; }
void stage_c_release(gp_job*) {}
int stage_c_run(gp_job*) { return GP_OK; }
}
