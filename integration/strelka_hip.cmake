# strelka-hip: the MI355X backend of Strelka's render library.
#
# Included by src/render/CMakeLists.txt (integration/strelka_hip.patch) when the Strelka tree is configured with
#     cmake -DSTRELKA_WITH_HIP=ON -DSTRELKA_HIP_DIR=<checkout of this repository> ...
# In place of the OptiX / CUDA sources it builds the `render` library from the reference's own render.cpp (RenderFactory, with the
# eCompute branch the patch adds) plus integration/HipRender.cpp, compiled against the REAL Strelka headers (-DSKH_WITH_STRELKA_HEADERS),
# and the device code as ONE shared library: libstrelka_hip.so (hipcc, gfx950; the flags of strelka_amd/build.py -- no fast-math, no fp
# contraction: the bit-exact contract with the CPU oracle depends on them).
#
# Variables it expects from the Strelka tree: ROOT_HOME, RENDERLIB_NAME, RENDER_SOURCES_COMMON, OUTPUT_DIRECTORY; targets: scene,
# materialmanager, glm::glm, stb::stb (found by the including file).

if(NOT STRELKA_HIP_DIR)
  message(FATAL_ERROR "STRELKA_WITH_HIP needs -DSTRELKA_HIP_DIR=<path to the strelka-hip repository>")
endif()
find_program(STRELKA_HIPCC hipcc HINTS /opt/rocm/bin ENV ROCM_PATH PATH_SUFFIXES bin REQUIRED)
set(STRELKA_HIP_ARCH "gfx950" CACHE STRING "offload architecture of the HIP kernels (MI355X = gfx950)")

set(SKH_SRC_DIR ${STRELKA_HIP_DIR}/strelka_amd/csrc)
set(SKH_LIBRARY ${OUTPUT_DIRECTORY}/libstrelka_hip.so)
add_custom_command(
  OUTPUT ${SKH_LIBRARY}
  COMMAND ${CMAKE_COMMAND} -E make_directory ${OUTPUT_DIRECTORY}
  COMMAND ${STRELKA_HIPCC} -O3 --offload-arch=${STRELKA_HIP_ARCH} -ffp-contract=off -fno-slp-vectorize
          -mllvm -amdgpu-sched-strategy=max-ilp -fPIC -shared -std=c++17
          -o ${SKH_LIBRARY} ${SKH_SRC_DIR}/strelka_hip.hip
  DEPENDS ${SKH_SRC_DIR}/strelka_hip.hip ${SKH_SRC_DIR}/skh_kernels.h ${SKH_SRC_DIR}/skh_device.h ${SKH_SRC_DIR}/skh_bvh.h
          ${STRELKA_HIP_DIR}/include/strelka_hip.h
  COMMENT "hipcc: libstrelka_hip.so (${STRELKA_HIP_ARCH})"
  VERBATIM)
add_custom_target(strelka_hip_kernels DEPENDS ${SKH_LIBRARY})

add_library(strelka_hip SHARED IMPORTED GLOBAL)
set_target_properties(strelka_hip PROPERTIES IMPORTED_LOCATION ${SKH_LIBRARY} IMPORTED_NO_SONAME TRUE)
add_dependencies(strelka_hip strelka_hip_kernels)

add_library(${RENDERLIB_NAME} STATIC
            ${RENDER_SOURCES_COMMON}
            ${STRELKA_HIP_DIR}/integration/HipRender.h
            ${STRELKA_HIP_DIR}/integration/HipRender.cpp
            ${STRELKA_HIP_DIR}/integration/SkhMaterials.h
            ${STRELKA_HIP_DIR}/integration/SkSceneDump.h)
# PUBLIC: HdStrelka and the apps see STRELKA_WITH_HIP (RenderDelegate.cpp asks the factory for eCompute; the .skscene exporter hooks)
target_compile_definitions(${RENDERLIB_NAME} PUBLIC STRELKA_WITH_HIP SKH_WITH_STRELKA_HEADERS)
target_include_directories(${RENDERLIB_NAME} PUBLIC ${STRELKA_HIP_DIR}/integration ${STRELKA_HIP_DIR}/include
                                                    ${ROOT_HOME}/include ${ROOT_HOME}/include/render ${ROOT_HOME})
target_link_libraries(${RENDERLIB_NAME} PUBLIC strelka_hip scene materialmanager glm::glm stb::stb)
set_target_properties(${RENDERLIB_NAME} PROPERTIES CXX_STANDARD 17 POSITION_INDEPENDENT_CODE ON)
