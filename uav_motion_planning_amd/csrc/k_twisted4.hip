// k_twisted4.hip -- solve_twisted_kernel<4, M, TILE, LPT> (qp_twisted.h) as its own translation unit (kernel_instances.h); no host code here.
#define UAVQP_KERNEL_TU
#include "qp_twisted.h"
#include "kernel_instances.h"
UAVQP_INSTANCES_TWISTED4
