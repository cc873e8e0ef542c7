// TEST SCAFFOLDING (tests/test_integration_files.py): stands where <settings/settings.h> of the Strelka tree would be found, so that the
// -DSKH_WITH_STRELKA_HEADERS branches of integration/*.{h,cpp} go through a compiler here (syntax and types only; nothing is linked or run).
// It forwards to this repository's own stand-in with the reference's SHAPE of Scene::MaterialDescription switched on.
#pragma once
#ifndef SKH_MIRROR_REFERENCE_MATERIALS
#    define SKH_MIRROR_REFERENCE_MATERIALS
#endif
#include "../../../../strelka_amd/host/oka_mirror.h"
