// Stand-in for <gflags/gflags.h> (InfiniTamDriver.h:11,19; InfiniTamDriver.cpp:6).  See tests/stubs/README.md.
#pragma once
#define DECLARE_bool(name) extern bool FLAGS_##name
#define DEFINE_bool(name, val, txt) bool FLAGS_##name = val
#define DECLARE_int32(name) extern int FLAGS_##name
#define DEFINE_int32(name, val, txt) int FLAGS_##name = val
#define DECLARE_string(name) extern std::string FLAGS_##name
#define DEFINE_string(name, val, txt) std::string FLAGS_##name = val
#define DECLARE_double(name) extern double FLAGS_##name
#define DEFINE_double(name, val, txt) double FLAGS_##name = val
