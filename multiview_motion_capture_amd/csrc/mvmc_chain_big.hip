// The BIG layout of the persistent chain kernel (config 5, C8 P8) as its own translation unit: see the note at the end of
// mvmc_chain.hip -- the out-of-line device functions take their register budget from the loosest kernel that reaches them, so the
// one-workgroup-per-CU BIG kernel must not share them with the three-workgroups-per-CU SMALL kernel.
#define MVMC_CHAIN_BIG_TU
#include "mvmc_chain.hip"
