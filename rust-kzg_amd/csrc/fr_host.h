// (host-side Fr helpers live in ckzg_internal.h / ff.cuh; this header is kept for the proving path)
#pragma once
#include "ff.cuh"
