// The single-pass level decoder a second time, with 4 KiB super-tiles: pqd::lv4k::k_levels behind
// pqd::launch_levels_4k (see pqd_device.h for when which build runs).
#define PQD_LV_TILE_BYTES 4096
#include "pqd_levels.hip"
