// k_corridor_dual.hip -- the kernels of qp_corridor_dual.h as their own translation unit (kernel_instances.h: UAVQP_INSTANCES_CORRIDOR_DUAL); no host code here.
#define UAVQP_KERNEL_TU
#include "qp_corridor_dual.h"
#include "kernel_instances.h"
UAVQP_INSTANCES_CORRIDOR_DUAL
