// k_rows32.hip -- rows_solve_kernel / rows_pair_kernel<3, 2, ...> (qp_rows.h, qp_rows2.h) as their own translation unit (kernel_instances.h); no host code here.
#define UAVQP_KERNEL_TU
#include "qp_rows2.h"
#include "kernel_instances.h"
UAVQP_INSTANCES_ROWS32
