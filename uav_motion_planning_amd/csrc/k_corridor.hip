// k_corridor.hip -- the kernels of qp_corridor.h as their own translation unit (kernel_instances.h: UAVQP_INSTANCES_CORRIDOR); no host code here.
#define UAVQP_KERNEL_TU
#include "qp_corridor.h"
#include "kernel_instances.h"
UAVQP_INSTANCES_CORRIDOR
