/* p7simd.h -- TEST INFRASTRUCTURE: the striped AVX2 forms of the byte MSV filter and the word Viterbi filter (oracle/p7simd.c), used by
 * oracle/p7oracle.c when p7o_set_simd(1) is in force.  Same tables, same results as the scalar restatement. */
#ifndef P7SIMD_H
#define P7SIMD_H
#include <stdint.h>
typedef struct P7S_PROF P7S_PROF;
/* rbv [29][M+1] biased byte costs; rwv [29][M+1] word scores; w8 [8][M+2] word transitions BM MM IM DM (into k) MD MI II DD (from k) */
P7S_PROF *p7s_create(int M, const uint8_t *rbv, int bias_b, int base_b, int tbm_b, int tec_b,
                     const int16_t *rwv, const int16_t *w8, int base_w, int wE_loop, int wE_move);
void p7s_free(P7S_PROF *s);
int p7s_msv(const P7S_PROF *s, const uint8_t *dsq, int L, int tjb_b, int *ret_xJ);      /* 0 ok, 1 overflow */
int p7s_vit(const P7S_PROF *s, const uint8_t *dsq, int L, int w_move, int *ret_xC);      /* 0 ok, 1 overflow */
#endif
