// k_rows_dual.hip -- the kernels of qp_rows_dual.h as their own translation unit (kernel_instances.h: UAVQP_INSTANCES_ROWS_DUAL); no host code here.
#define UAVQP_KERNEL_TU
#include "qp_rows_dual.h"
#include "kernel_instances.h"
UAVQP_INSTANCES_ROWS_DUAL
