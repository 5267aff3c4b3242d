// Wave-shift disc kernels: the per-size specialisations (every odd disc size from 3 to 101) are split over TOPO_NGROUPS
// translation units so that they compile in parallel - this ONE source, compiled once per group with -DTOPO_GROUP=<k>
// (build.py).  Group k holds the sizes with ((size - 3) / 2) % TOPO_NGROUPS == k.
#include "disc_wave_impl.hpp"

#ifndef TOPO_GROUP
#error "compile with -DTOPO_GROUP=<k> (topo_descriptors_amd/build.py does)"
#endif
#ifndef TOPO_NGROUPS
#define TOPO_NGROUPS 16
#endif

namespace topo {

namespace {
template <int SIZE, bool MINE>
struct GroupCall {
    static int go(const Block&, float*, float*) { return TOPO_AMD_EUNSUP; }
};
template <int SIZE>
struct GroupCall<SIZE, true> {  // (only the sizes of this group instantiate their kernels)
    static int go(const Block& b, float* tpi_out, float* std_out) { return launch_wave_any<SIZE>(b, tpi_out, std_out); }
};
}  // namespace

#define TOPO_GROUP_NAME_(g) launch_disc_wave_group##g
#define TOPO_GROUP_NAME(g) TOPO_GROUP_NAME_(g)

int TOPO_GROUP_NAME(TOPO_GROUP)(const Block& b, int size, float* tpi_out, float* std_out) {
    switch (size) {
#define TOPO_SIZE_CASE(S) \
    case S: return GroupCall<S, ((S - 3) / 2) % TOPO_NGROUPS == TOPO_GROUP>::go(b, tpi_out, std_out);
        TOPO_SIZE_CASE(3) TOPO_SIZE_CASE(5) TOPO_SIZE_CASE(7) TOPO_SIZE_CASE(9) TOPO_SIZE_CASE(11) TOPO_SIZE_CASE(13) TOPO_SIZE_CASE(15)
        TOPO_SIZE_CASE(17) TOPO_SIZE_CASE(19) TOPO_SIZE_CASE(21) TOPO_SIZE_CASE(23) TOPO_SIZE_CASE(25) TOPO_SIZE_CASE(27) TOPO_SIZE_CASE(29)
        TOPO_SIZE_CASE(31) TOPO_SIZE_CASE(33) TOPO_SIZE_CASE(35) TOPO_SIZE_CASE(37) TOPO_SIZE_CASE(39) TOPO_SIZE_CASE(41) TOPO_SIZE_CASE(43)
        TOPO_SIZE_CASE(45) TOPO_SIZE_CASE(47) TOPO_SIZE_CASE(49) TOPO_SIZE_CASE(51) TOPO_SIZE_CASE(53) TOPO_SIZE_CASE(55) TOPO_SIZE_CASE(57)
        TOPO_SIZE_CASE(59) TOPO_SIZE_CASE(61) TOPO_SIZE_CASE(63) TOPO_SIZE_CASE(65) TOPO_SIZE_CASE(67) TOPO_SIZE_CASE(69) TOPO_SIZE_CASE(71)
        TOPO_SIZE_CASE(73) TOPO_SIZE_CASE(75) TOPO_SIZE_CASE(77) TOPO_SIZE_CASE(79) TOPO_SIZE_CASE(81) TOPO_SIZE_CASE(83) TOPO_SIZE_CASE(85)
        TOPO_SIZE_CASE(87) TOPO_SIZE_CASE(89) TOPO_SIZE_CASE(91) TOPO_SIZE_CASE(93) TOPO_SIZE_CASE(95) TOPO_SIZE_CASE(97) TOPO_SIZE_CASE(99)
        TOPO_SIZE_CASE(101)
#undef TOPO_SIZE_CASE
        default: return TOPO_AMD_EUNSUP;
    }
}

}  // namespace topo
