// counter access for the operation-counting build (see mjo_count.h); compiled WITHOUT the `double` redefinition
thread_local unsigned long long mjo_nflop = 0;
extern "C" unsigned long long mjo_flops_get(void) { return mjo_nflop; }
extern "C" void mjo_flops_reset(void) { mjo_nflop = 0; }
