// k_generic.hip -- the kernels of qp_generic2.h as their own translation unit (kernel_instances.h: UAVQP_INSTANCES_GENERIC); no host code here.
#define UAVQP_KERNEL_TU
#include "qp_generic2.h"
#include "kernel_instances.h"
UAVQP_INSTANCES_GENERIC
