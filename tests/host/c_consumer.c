/* A consumer of include/rapflow.h written in plain C99 (tests/test_abi.py compiles it with gcc -std=c99 -pedantic, links it against
 * librapflow.so and runs it): what a non-Python binding of the boundary sees.  No GPU needed -- every call below is answered by the
 * host side of the library (version, workspace arithmetic, argument validation). */
#include <stdio.h>
#include <stddef.h>
#include "rapflow.h"

int main(void) {
  int fails = 0;
  char sentinel[256];
  if (rap_version() != RAPFLOW_ABI_VERSION) { printf("version %d != header %d\n", rap_version(), RAPFLOW_ABI_VERSION); ++fails; }
  if (rap_attention_workspace_bytes(262144, 64) == 0) { printf("attention workspace query\n"); ++fails; }
  if (rap_rigidity_workspace_bytes(64, 20, 32) == 0 || rap_rigidity_workspace_bytes(-1, 0, 0) != 0) { printf("rigidity workspace query\n"); ++fails; }
  if (rap_set_tuning(-1, 0) != RAP_ERR_INVALID) { printf("tuning key -1 accepted\n"); ++fails; }
  /* NULL operands and bad strides are refused before any launch */
  if (rap_x2_gemm(1, NULL, 1024, (const uint16_t*)sentinel, 1024, sentinel, 512, 256, 512, 1024, NULL, NULL, 0, 1.0f, 0, NULL, NULL, 8.0f, NULL, 0, NULL) != RAP_ERR_INVALID) { printf("x2 gemm NULL A\n"); ++fails; }
  if (rap_x2_gemm(1, (const uint16_t*)sentinel, 1024, (const uint16_t*)sentinel, 1024, sentinel, 514, 256, 512, 1024, NULL, NULL, 0, 1.0f, 0, NULL, NULL, 8.0f, NULL, 0, NULL) != RAP_ERR_INVALID) { printf("x2 gemm ldc 514\n"); ++fails; }
  if (rap_model_set_compute_dtype(NULL, 3, NULL) != RAP_ERR_INVALID) { printf("NULL model\n"); ++fails; }
  printf("c consumer: %d failure(s), ABI version %d\n", fails, rap_version());
  return fails;
}
