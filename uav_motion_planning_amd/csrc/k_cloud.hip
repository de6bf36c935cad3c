// k_cloud.hip -- the kernels of obstacle_grid.h as their own translation unit (kernel_instances.h: UAVQP_INSTANCES_CLOUD); no host code here.
#define UAVQP_KERNEL_TU
#include "obstacle_grid.h"
#include "kernel_instances.h"
UAVQP_INSTANCES_CLOUD
