// The SMALL layout of the persistent chain kernel built for 256 VGPRs (two workgroups per CU) as its own translation unit: the LATENCY
// build, which mvmc_chain_run launches when the call has few workgroups (MvTracker.update_4d: one chain of one frame).  The throughput
// build (mvmc_chain.hip: 128 VGPRs, four workgroups per CU) pays for its occupancy with smaller batches and 888 bytes of scratch per
// lane; this one uses the batch sizes of the 168-register build with room to spare (544 bytes of scratch).  Shelf through update_4d,
// same box: 600 frames/s (128) -> 648 (168) -> 669 (256; 512 allowed: 334 used, 667).  Same results bit for bit.
#define MVMC_SMALL_WPS 2
#define MVMC_CHAIN_LAT_TU
#include "mvmc_chain.hip"
