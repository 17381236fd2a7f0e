# Builds libautogp_hip.so for gfx950 next to the HIP sources (the same command as __graft_entry__.build()) and records
# its path for src/AutoGPHIP.jl.  Never executed here (no Julia in the build image).
const CSRC = normpath(joinpath(@__DIR__, "..", "..", "csrc"))
const LIBDIR = normpath(joinpath(@__DIR__, "..", "..", "lib"))
const LIBFILE = joinpath(LIBDIR, "libautogp_hip.so")

hipcc = something(Sys.which("hipcc"), "/opt/rocm/bin/hipcc")
isfile(hipcc) || error("hipcc not found: the engine has no CPU path, a ROCm toolchain is required")
mkpath(LIBDIR)
run(Cmd(`$hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o $LIBFILE agp_engine.hip -ldl`; dir=CSRC))
open(joinpath(@__DIR__, "deps.jl"), "w") do io
    println(io, "const libautogp_hip = ", repr(LIBFILE))
end
