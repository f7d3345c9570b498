// Test hook: leave a chosen bit pattern in every CU's LDS and vector registers, so that a test can show that a
// kernel's results do not depend on what ran before it (on-chip state is NOT cleared between launches: a read of an
// LDS word or a register the kernel never wrote returns whatever the previous workgroup left there).
#include "common.h"

namespace fastgrnn {
namespace {

// one workgroup per CU at a time (the whole 160 KB of LDS), 256 threads = one wave per SIMD = all 512 registers
__global__ __launch_bounds__(256) void poison_cu_state(unsigned pattern, unsigned* sink) {
  extern __shared__ unsigned lds[];
  for (unsigned k = threadIdx.x; k < 160u * 1024u / 4u; k += 256u) lds[k] = pattern;
  __syncthreads();
  unsigned v = pattern;
#define P8(b) "v_mov_b32 v" #b "0, %0\n v_mov_b32 v" #b "1, %0\n v_mov_b32 v" #b "2, %0\n v_mov_b32 v" #b "3, %0\n v_mov_b32 v" #b "4, %0\n" \
              " v_mov_b32 v" #b "5, %0\n v_mov_b32 v" #b "6, %0\n v_mov_b32 v" #b "7, %0\n v_mov_b32 v" #b "8, %0\n v_mov_b32 v" #b "9, %0\n"
#define A8(b) "v_accvgpr_write_b32 a" #b "0, %0\n v_accvgpr_write_b32 a" #b "1, %0\n v_accvgpr_write_b32 a" #b "2, %0\n v_accvgpr_write_b32 a" #b "3, %0\n" \
              " v_accvgpr_write_b32 a" #b "4, %0\n v_accvgpr_write_b32 a" #b "5, %0\n v_accvgpr_write_b32 a" #b "6, %0\n v_accvgpr_write_b32 a" #b "7, %0\n" \
              " v_accvgpr_write_b32 a" #b "8, %0\n v_accvgpr_write_b32 a" #b "9, %0\n"
  // v10 .. v249 and a10 .. a249 (the compiler keeps its own few values below v10)
  asm volatile(P8(1) P8(2) P8(3) P8(4) P8(5) P8(6) P8(7) P8(8) P8(9) P8(10) P8(11) P8(12) P8(13) P8(14) P8(15) P8(16) P8(17) P8(18)
               P8(19) P8(20) P8(21) P8(22) P8(23) P8(24)
               A8(1) A8(2) A8(3) A8(4) A8(5) A8(6) A8(7) A8(8) A8(9) A8(10) A8(11) A8(12) A8(13) A8(14) A8(15) A8(16) A8(17) A8(18)
               A8(19) A8(20) A8(21) A8(22) A8(23) A8(24)
               :: "v"(v)
               : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27",
                 "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45",
                 "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63",
                 "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81",
                 "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99",
                 "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114",
                 "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129",
                 "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144",
                 "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159",
                 "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174",
                 "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189",
                 "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204",
                 "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219",
                 "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234",
                 "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249",
                 "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27",
                 "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45",
                 "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63",
                 "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81",
                 "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99",
                 "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114",
                 "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129",
                 "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144",
                 "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159",
                 "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174",
                 "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189",
                 "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204",
                 "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219",
                 "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234",
                 "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249");
#undef P8
#undef A8
  if (lds[(threadIdx.x * 97u) % (160u * 1024u / 4u)] == 0x12345u && sink) sink[0] = 1u;   // keeps the LDS writes
}

}  // namespace

int debug_poison(unsigned pattern, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(poison_cu_state), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess) return FASTGRNN_ERR_LAUNCH;
    attr_set = true;
  }
  // several workgroups per CU in turn: every CU is visited however the dispatcher places them
  hipLaunchKernelGGL(poison_cu_state, dim3(256 * 8), dim3(256), 160 * 1024, s, pattern, (unsigned*)nullptr);
  return hipGetLastError() == hipSuccess ? FASTGRNN_OK : FASTGRNN_ERR_LAUNCH;
}

}  // namespace fastgrnn
