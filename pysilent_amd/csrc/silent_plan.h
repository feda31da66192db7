// silent_pyramid_plan: the host object behind silent_pyramid_plan_create (silent_pyramid_api.hip builds it; the gray pass of
// silent_gray_api.hip reads its unit levels and stream tables).
#pragma once

#include "silent_internal.h"
#include "silent_gray.h"
#include "silent_pyramid.h"
#include "silent_walk_rgb.h"

struct silent_pyramid_plan {
    silent_ctx* ctx = nullptr;
    silent::PyrTab tab{};
    std::vector<silent_extent> extents;
    void* tables = nullptr;
    float unit_w[6] = {0, 0, 0, 0, 0, 0};  // scipy's six taps of a unit-zoom level ([1,26,66,26,1]/120 and 2^-53, as float32)
    // single-read "stream" path (gray_stream_kernel): row programs + column records, when the plan is eligible
    bool stream_ok = false;
    void* stream_tables = nullptr;
    silent::StreamTab stream{};
    int stream_unit_level = -1;
    int stream_layout = 0;               // slot layout of the row programs (stream_slots, silent_gray.h): 1 = dense ladders, 7-level kernels
    // walk plans of pyramid_walk3_kernel (silent_walk_rgb.h; 3 channels): a classic pyramid is ONE plan (unit level + every
    // other level on the same crop), a crop layout like the reference's one plan per level; row programs (completion records)
    // + column records per wave tile live in walk_tables
    bool walk_pyr_ok = false;
    int walk_px = 36;                    // pixels per consumer wave: 36; 32 / 28 / 24 for zoom steps below 1.875 / 1.6 / 1.4
    int walk_G = 4;                      // general levels the kernel is instantiated for (4 or 7)
    void* walk_tables = nullptr;
    silent::Walk3Args walk{};                    // everything but the per-launch decomposition (strips / segments / block0)
    silent::BorderTab walk_border{};             // union plans: the inner levels' border outputs (pyramid_border_kernel); n = 0: none
};

// (silent_pyramid_api.hip) with_unit: also the unit levels (the gray pass produces them itself); with_region: also the general levels
int launch_pyramid(silent_ctx* ctx, const char* who, const silent_pyramid_plan* plan, const float* frames, int n_frames, float* pyr,
                   hipStream_t s, bool with_unit, bool with_region = true);
