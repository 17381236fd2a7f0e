# Builds libautogp_hip.so for gfx950 from ALL translation units of csrc/ (csrc/Makefile: the recipe __graft_entry__.build()
# runs as well) and records its path for src/AutoGPHIP.jl.  Never executed here (no Julia in the build image); the command
# below is extracted and run by tests/test_host.py::test_documented_build_recipes_produce_the_full_library.
const CSRC = normpath(joinpath(@__DIR__, "..", "..", "csrc"))
const LIBDIR = normpath(joinpath(@__DIR__, "..", "..", "lib"))
const OBJDIR = normpath(joinpath(@__DIR__, "..", "..", "build", "obj"))
const LIBFILE = joinpath(LIBDIR, "libautogp_hip.so")

hipcc = something(Sys.which("hipcc"), "/opt/rocm/bin/hipcc")
isfile(hipcc) || error("hipcc not found: the engine has no CPU path, a ROCm toolchain is required")
mkpath(LIBDIR)
run(`make -C $CSRC -j8 HIPCC=$hipcc OUT=$LIBFILE OBJDIR=$OBJDIR`)
open(joinpath(@__DIR__, "deps.jl"), "w") do io
    println(io, "const libautogp_hip = ", repr(LIBFILE))
end
