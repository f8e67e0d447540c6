// Stub for the two enums Renderer/Texture.h needs; no CUDA is involved.
#pragma once
enum CUarray_format { CU_AD_FORMAT_UNSIGNED_INT8, CU_AD_FORMAT_UNSIGNED_INT32 };
enum CUresourceViewFormat { CU_RES_VIEW_FORMAT_UINT_4X8, CU_RES_VIEW_FORMAT_UNSIGNED_BC1, CU_RES_VIEW_FORMAT_UNSIGNED_BC2, CU_RES_VIEW_FORMAT_UNSIGNED_BC3 };
