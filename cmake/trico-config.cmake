# trico-config.cmake — lets a CMake consumer of the reference (target `trico`, CMakeLists.txt:19-20 and
# trico/CMakeLists.txt:27-47 of janm31415/trico; `trico_io` for the STL / PLY readers) switch to this library:
#
#     find_package(trico CONFIG REQUIRED PATHS /path/to/this/repo/cmake)
#     target_link_libraries(app PRIVATE trico::trico)            # or trico::trico_io
#
# The libraries are built in-tree by `python -m trico_amd.build` (hipcc, gfx950); this file only describes them.
get_filename_component(_trico_root "${CMAKE_CURRENT_LIST_DIR}/.." ABSOLUTE)
set(trico_INCLUDE_DIR "${_trico_root}/include")
set(trico_LIBRARY_DIR "${_trico_root}/trico_amd/lib")

if(NOT EXISTS "${trico_LIBRARY_DIR}/libtrico.so")
  set(trico_FOUND FALSE)
  set(trico_NOT_FOUND_MESSAGE "libtrico.so has not been built: run `python -m trico_amd.build` in ${_trico_root}")
  return()
endif()

if(NOT TARGET trico::trico)
  add_library(trico::trico SHARED IMPORTED)
  set_target_properties(trico::trico PROPERTIES
    IMPORTED_LOCATION "${trico_LIBRARY_DIR}/libtrico.so"
    IMPORTED_NO_SONAME TRUE
    INTERFACE_INCLUDE_DIRECTORIES "${trico_INCLUDE_DIR}")
endif()
# the static flavour (the reference's default, trico/CMakeLists.txt:27-34: TRICO_SHARED off): libtrico.a needs the HIP runtime and the C++
# runtime of its HIP objects at link time.  `set(TRICO_SHARED OFF)` before find_package makes the plain name `trico` mean this one.
if(NOT TARGET trico::trico_static AND EXISTS "${trico_LIBRARY_DIR}/libtrico.a")
  add_library(trico::trico_static STATIC IMPORTED)
  find_library(_trico_hip amdhip64 HINTS /opt/rocm/lib ENV ROCM_PATH PATH_SUFFIXES lib)
  set_target_properties(trico::trico_static PROPERTIES
    IMPORTED_LOCATION "${trico_LIBRARY_DIR}/libtrico.a"
    INTERFACE_INCLUDE_DIRECTORIES "${trico_INCLUDE_DIR}"
    INTERFACE_LINK_LIBRARIES "${_trico_hip};stdc++;m;dl;pthread")
endif()
if(NOT TARGET trico::trico_io AND EXISTS "${trico_LIBRARY_DIR}/libtrico_io.so")
  add_library(trico::trico_io SHARED IMPORTED)
  set_target_properties(trico::trico_io PROPERTIES
    IMPORTED_LOCATION "${trico_LIBRARY_DIR}/libtrico_io.so"
    IMPORTED_NO_SONAME TRUE
    INTERFACE_INCLUDE_DIRECTORIES "${trico_INCLUDE_DIR}"
    INTERFACE_LINK_LIBRARIES trico::trico)
endif()
# the reference's plain target names, for CMakeLists that say target_link_libraries(app trico)
if(NOT TARGET trico)
  add_library(trico INTERFACE IMPORTED)
  if(DEFINED TRICO_SHARED AND NOT TRICO_SHARED AND TARGET trico::trico_static)
    set_target_properties(trico PROPERTIES INTERFACE_LINK_LIBRARIES trico::trico_static)
  else()
    set_target_properties(trico PROPERTIES INTERFACE_LINK_LIBRARIES trico::trico)
  endif()
endif()
if(NOT TARGET trico_io AND TARGET trico::trico_io)
  add_library(trico_io INTERFACE IMPORTED)
  set_target_properties(trico_io PROPERTIES INTERFACE_LINK_LIBRARIES trico::trico_io)
endif()
set(trico_FOUND TRUE)
