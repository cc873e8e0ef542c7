// Convenience header of this repository's own host build: the stand-in types (oka_mirror.h) + the backend (integration/HipRender.h).
// A Strelka tree includes integration/HipRender.h directly, with -DSKH_WITH_STRELKA_HEADERS.
#pragma once
#include "oka_mirror.h"
#include "../../integration/HipRender.h"
