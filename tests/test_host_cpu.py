"""CPU tests (-m "not gpu") of the host-side pieces: the C ABI exports every declared symbol, the parameter-file reader
accepts the reference's file grammar, the stream sharding + timing reduction work across 2 processes (gloo)."""
import ctypes as C
import os
import re
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"BF_API\s+[\w\s\*]+?\b(bf_\w+)\s*\(", txt)))


@pytest.mark.parametrize("header", ["bf_hip.h", "bf_pipeline.h", "bf_sensordata.h"])
def test_library_exports_every_declared_symbol(built, header):
    lib = C.CDLL(os.path.join(ROOT, "bundlefusion_amd", "lib", "libbf_hip.so"))
    names = _declared(header)
    assert len(names) > {"bf_hip.h": 60, "bf_pipeline.h": 80, "bf_sensordata.h": 13}[header]
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_no_oracle_symbols_in_product(built):
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "bundlefusion_amd", "lib", "libbf_hip.so")]).decode()
    assert " or_" not in out and "oracle" not in out.lower()
    for src in os.listdir(os.path.join(ROOT, "bundlefusion_amd", "csrc")):
        body = open(os.path.join(ROOT, "bundlefusion_amd", "csrc", src)).read()
        assert "oracle/" not in body and "or_common.h" not in body, src


def test_calls_fail_loudly_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bundlefusion_amd.capi import lib
    h = C.c_void_p()
    rc = lib.bf_siftmgr_create(4, 64, C.byref(h))
    assert rc != 0 and lib.bf_last_error()          # no silent CPU fallback


def test_parameter_files(built, tmp_path):
    from bundlefusion_amd.capi import lib, GlobalAppState, GlobalBundlingState, default_app_state, default_bundling_state
    app = tmp_path / "zParametersDefault.txt"
    app.write_text(textwrap.dedent('''
        // 0=Kinect; 8=SensorDataReader (for offline processing)
        s_sensorIdx = 8;
        s_numSolveFramesBeforeExit = -1;//30 //#frames to run after solve done
        s_generateVideoDir = "output/";   // unknown / rendering keys are ignored
        s_topVideoTransformWorld = 1.0f 0.0f 0.0f 0.0f 0.0f 1.0f 0.0f 0.0f 0.0f 0.0f 1.0f 0.0f 0.0f 0.0f 0.0f 1.0f;
        s_integrationWidth = 640;	//input depth gets re-sampled to this width
        s_integrationHeight = 480;
        s_maxFrameFixes = 7;
        s_SDFVoxelSize = 0.004f;				//voxel size in meter
        s_SDFTruncation = 0.06f;
        s_hashNumBuckets = 2000000;
        s_streamingGridDimensions = 257 257 129; // dimensions have to be odd
        s_streamingVoxelExtents = 1.0f 2.0f 0.5f;
        s_binaryDumpSensorFile = "../data/se//quence.sens";
        s_colorFilter = true;
        s_garbageCollectionEnabled	= false;
    '''))
    g = GlobalAppState(); miss = C.c_uint32()
    assert lib.bf_global_app_state_read(str(app).encode(), C.byref(g), C.byref(miss)) == 0
    assert (g.s_sensorIdx, g.s_integrationWidth, g.s_integrationHeight, g.s_maxFrameFixes, g.s_hashNumBuckets) == (8, 640, 480, 7, 2000000)
    assert g.s_numSolveFramesBeforeExit == 0xFFFFFFFF                      # -1
    assert abs(g.s_SDFVoxelSize - 0.004) < 1e-9 and abs(g.s_SDFTruncation - 0.06) < 1e-8
    assert list(g.s_streamingGridDimensions) == [257, 257, 129] and list(g.s_streamingVoxelExtents) == [1.0, 2.0, 0.5]
    assert g.s_colorFilter == 1 and g.s_garbageCollectionEnabled == 0
    d = default_app_state()
    assert g.s_topNActive == d.s_topNActive == 30 and g.s_hashNumSDFBlocks == 200000 and miss.value > 5        # untouched keys keep the shipped defaults
    bnd = tmp_path / "zParametersBundlingDefault.txt"
    bnd.write_text("s_submapSize = 5;\ns_minKeyScale = 5.0f;//3.0f\ns_useLocalVerify = false;\n//s_downsampledWidth = 160;\ns_downsampledWidth = 80;\n")
    b = GlobalBundlingState()
    assert lib.bf_global_bundling_state_read(str(bnd).encode(), C.byref(b), None) == 0
    assert b.s_submapSize == 5 and b.s_minKeyScale == 5.0 and b.s_useLocalVerify == 0 and b.s_downsampledWidth == 80
    db = default_bundling_state()
    assert (db.s_maxNumImages, db.s_submapSize, db.s_numLocalNonLinIterations, db.s_numGlobalLinIterations) == (1200, 10, 2, 150)
    assert lib.bf_global_app_state_read(b"/nonexistent/file.txt", C.byref(g), None) != 0 and b"cannot open" in lib.bf_last_error()


def test_reference_parameter_files_if_present(built):
    ref = "/root/reference/FriedLiver"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not mounted")
    from bundlefusion_amd.capi import lib, GlobalAppState, GlobalBundlingState, default_app_state, default_bundling_state
    g = GlobalAppState(); b = GlobalBundlingState(); m1 = C.c_uint32(); m2 = C.c_uint32()
    assert lib.bf_global_app_state_read((ref + "/zParametersDefault.txt").encode(), C.byref(g), C.byref(m1)) == 0
    assert lib.bf_global_bundling_state_read((ref + "/zParametersBundlingDefault.txt").encode(), C.byref(b), C.byref(m2)) == 0
    assert m1.value == 0 and m2.value == 0                                  # every field the hot path reads is set by the shipped files
    assert bytes(g) == bytes(default_app_state()) and bytes(b) == bytes(default_bundling_state())     # the built-in defaults ARE the shipped files


def test_segments_are_disjoint_and_cover():
    from bundlefusion_amd.shard import segment
    world, per = 8, 210
    segs = [segment(r, world, per) for r in range(world)]
    assert segs[0] == (0, 210) and all(segs[i][1] == segs[i + 1][0] for i in range(world - 1)) and segs[-1][1] == world * per
    with pytest.raises(ValueError):
        segment(8, 8, 10)


_WORKER = '''
import os, sys, time
sys.path.insert(0, %r)
import torch.distributed as dist
import torch
from bundlefusion_amd.shard import segment, max_over_ranks, whole_job_rate, same_over_ranks
dist.init_process_group("gloo")
r, w = dist.get_rank(), dist.get_world_size()
first, last = segment(r, w, 20)
elapsed = 0.5 + r                      # rank 1 is the slow one
dist.barrier()
mx = max_over_ranks(elapsed)
rate = whole_job_rate(last - first, elapsed)
same = same_over_ranks(torch.arange(12, dtype=torch.float32).reshape(3, 4) * 0.37)
diff = same_over_ranks(torch.full((4,), 1.0 + 1e-7 * r))
os.write(1, ("RESULT {} {} {} {!r} {!r} {} {}\\n".format(r, first, last, mx, rate, int(same), int(diff))).encode())   # one write: lines of two ranks never interleave
dist.destroy_process_group()
'''


def test_two_rank_gloo_sharding_and_timing(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % ROOT)
    import socket
    with socket.socket() as sock:                      # a free port: a fixed one collides with a concurrent run of the suite
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", port, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = sorted(l.split()[1:] for l in out.stdout.splitlines() if l.startswith("RESULT"))
    assert len(rows) == 2
    assert [(int(r[1]), int(r[2])) for r in rows] == [(0, 20), (20, 40)]
    assert all(abs(float(r[3]) - 1.5) < 1e-9 for r in rows)                 # MAX over ranks
    assert all(abs(float(r[4]) - 2 * 20 / 1.5) < 1e-9 for r in rows)        # whole-job units / slowest rank
    assert all(r[5] == "1" and r[6] == "0" for r in rows)                   # bit-identical tensors pass, a 1-ulp difference is caught


_CHUNK_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import torch.distributed as dist
from bundlefusion_amd.shard import ChunkedRunner, chunk_owner, chunk_frames

dist.init_process_group("gloo")
r, w = dist.get_rank(), dist.get_world_size()
S = 10


class FakeWorker:                      # a package = 64 bytes tagged with (producing rank, chunk, first frame, last frame)
    package_bytes = 64
    def __init__(self): self.ran = []
    def run(self, chunk, frames, out=None):
        assert len(frames) == S + 1 and frames[0] == chunk * S and frames[-1] == chunk * S + S      # the chunk's frames, first one shared
        out[:] = 0
        out[:16].view(np.int32)[:] = [r, chunk, frames[0], frames[-1]]
        self.ran.append(chunk)
        return out


class FakePipe:
    def __init__(self): self.log = []
    def process_frame_chunked(self, d, c, pkg, j):
        tag = pkg[:16].view(np.int32)
        self.log.append((int(d), int(tag[0]), int(tag[1]), j))
        return True


class Frame(int):
    pass


n = 1 + 5 * S                                  # 5 local chunks over 2 ranks: the last round is half empty
runner = ChunkedRunner(FakePipe(), FakeWorker(), None, S, r, w, device="cpu")
feed = [(Frame(i), Frame(i)) for i in range(runner.frames_needed(n))]
runner.feed = feed
# the worker of this test takes frame numbers: hand it the first element of each pair
orig_run = runner.worker.run
runner.worker.run = lambda chunk, frames, out=None: orig_run(chunk, [int(f[0]) for f in frames], out)
runner.advance(7); runner.advance(n - 7)       # any split of the stream gives the same schedule
log = runner.pipe.log
ok_order = [f for f, _, _, _ in log] == list(range(n))
ok_owner = all(prod == chunk_owner(c, w) and c == (0 if f == 0 else (f - 1) // S) and j == f - c * S for f, prod, c, j in log)
os.write(1, ("CHUNKS {} {} {} {} {} {}\n".format(r, int(ok_order), int(ok_owner), ",".join(map(str, runner.worker.ran)), runner.rounds, len(feed))).encode())
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_chunk_parallel_schedule(tmp_path, world):
    """Chunk-parallel mode across 2 and 4 processes (gloo, CPU): local chunks go round robin to the ranks, one all-gather per round of
    `world` chunks delivers every package to every rank in owner order, and every rank then walks ALL frames in stream order with
    the package of the frame's chunk (fake worker / pipeline objects record the schedule)."""
    script = tmp_path / "chunk_worker.py"
    script.write_text(_CHUNK_WORKER % ROOT)
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
                          "--master-port", port, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = sorted(l.split()[1:] for l in out.stdout.splitlines() if l.startswith("CHUNKS"))
    assert len(rows) == world
    assert all(r[1] == "1" and r[2] == "1" for r in rows)                    # every frame once, in order, with the right package and local index
    rounds = -(-5 // world)                                                  # 5 local chunks of the stream; the last round may reach beyond them
    for rank, r in enumerate(rows):
        assert r[3] == ",".join(str(rank + k * world) for k in range(rounds))      # round robin inside rounds of `world` chunks
        assert r[4] == str(rounds) and r[5] == str(rounds * world * 10 + 1)  # one collective per round; the stream is extended to complete the last round


def test_cpp_header_classes_compile_and_link(built, tmp_path):
    """include/bundlefusion/bundlefusion.hpp (reference class names over the C ABI) builds with plain g++ — no HIP headers
    needed on the integrator's side — and every forwarded symbol resolves against libbf_hip.so."""
    libdir = os.path.join(ROOT, "bundlefusion_amd", "lib")
    for name in ("headless_driver", "sens_pipeline"):
        exe = tmp_path / name
        r = subprocess.run(["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", name + ".cpp"),
                            "-L", libdir, "-lbf_hip", "-Wl,-rpath," + libdir, "-o", str(exe)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and "usage:" in out.stdout
    import torch
    if not torch.cuda.is_available():            # the C++ .sens player reads the file on the host and then fails loudly at the first device call
        from bundlefusion_amd import sensordata as sdm
        with sdm.SensorDataWriter(tmp_path / "t.sens", (16, 12), (16, 12), np.eye(4, dtype=np.float32)) as w:
            w.add_frame(np.eye(4, dtype=np.float32), np.full((12, 16), 1500, np.uint16), bytes(16 * 12 * 3))
        out = subprocess.run([str(tmp_path / "sens_pipeline"), str(tmp_path / "t.sens")], capture_output=True, text=True, timeout=120)
        assert out.returncode == 1 and "error: bundlefusion:" in out.stdout and "hip" in out.stdout.lower()


def test_integrate_colour_rounding_shortcut_is_exact():
    """tsdf.hip voxelApply<integrate> rounds 0.2f*c + 0.8f*o with round-to-nearest-even (v_rndne) where the reference uses
    roundf (half away from zero, VoxelUtilHashSDF.h combineVoxel): identical for every pair of bytes, in float32 arithmetic."""
    c, o = np.meshgrid(np.arange(256, dtype=np.float32), np.arange(256, dtype=np.float32), indexing="ij")
    r = (np.float32(0.2) * c).astype(np.float32) + (np.float32(0.8) * o).astype(np.float32)
    assert r.dtype == np.float32
    half_away = np.copysign(np.floor(np.abs(r) + np.float32(0.5)), r)            # roundf
    assert np.array_equal(np.rint(r), half_away)
    frac = np.abs(r - np.floor(r) - 0.5)
    assert frac.min() > 0.09                                                      # never near a tie: (c + 4 o) / 5


_TIMING_CPP = r'''
#include "bundlefusion/bundlefusion.hpp"
using namespace bundlefusion;
int main(int argc, char** argv) {
    TimingLog::init();
    for (int k = 0; k < 3; ++k) {
        bf_frame_timing t; std::memset(&t, 0, sizeof t);
        t.timeSensorProcess = 0.25f * (k + 1); t.timeSiftDetection = 1.5f; t.timeSiftMatching = 0.5f; t.timeMatchFilter = 0.125f * k; t.timeSolve = 2.0f * k;
        t.timeReIntegrate = 1.0f; t.timeReconstruct = 0.75f; t.timeTotal = 4.0f + k;
        TimingLog::addLocalFrameTiming(t);
    }
    TimingLog::addGlobalFrameTiming();
    TimingLog::getFrameTiming(false).timeSolve = 12.5; TimingLog::getFrameTiming(false).numItersSolve = 3;
    TimingLog::printAllTimings(std::string(argv[1]) + "/");
    {   // mat4f: product and general inverse (what g_transformWorld * transformation / getInverse() need)
        mat4f A = mat4f::identity();
        const float v[16] = {0.36f, 0.48f, -0.8f, 1.5f, -0.8f, 0.6f, 0.0f, -2.0f, 0.48f, 0.64f, 0.6f, 0.25f, 0, 0, 0, 1};
        for (int i = 0; i < 16; ++i) A.m[i] = v[i];
        mat4f K = mat4f::identity(); K(0, 0) = 583; K(1, 1) = 584; K(0, 2) = 319.5f; K(1, 2) = 239.5f;
        const mat4f P = A * A.getInverse(), Q = K.getInverse() * K;
        for (int i = 0; i < 16; ++i) { const float id = (i % 5 == 0) ? 1.0f : 0.0f; if (P.m[i] - id > 1e-5f || id - P.m[i] > 1e-5f || Q.m[i] - id > 1e-5f || id - Q.m[i] > 1e-5f) return 4; }
    }
    // SensorDataReader over a file that does not exist: the C ABI's message surfaces as an exception
    try { SensorDataReader r; r.createFirstConnected(std::string(argv[1]) + "/missing.sens"); return 2; }
    catch (const std::exception& e) { if (std::string(e.what()).find("could not open") == std::string::npos) return 3; }
    return 0;
}
'''


def test_timing_log_writes_the_reference_formats(built, tmp_path):
    """TimingLog (TimingLog.h:39-233) in bundlefusion.hpp: the per-frame text log and the comma-separated "excel" files."""
    src = tmp_path / "tl.cpp"
    src.write_text(_TIMING_CPP)
    exe = tmp_path / "tl"
    libdir = os.path.join(ROOT, "bundlefusion_amd", "lib")
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), str(src), "-L", libdir, "-lbf_hip", "-Wl,-rpath," + libdir,
                        "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = tmp_path / "timings"
    out.mkdir()
    assert subprocess.run([str(exe), str(out)], capture_output=True, text=True, timeout=60).returncode == 0
    log = (out / "timingLogPerFrame.txt").read_text()
    assert log.startswith("Global Timings Per Frame:\n[ frame 0 ]\n\tTime SIFT Detection: 0.000000ms\n")
    assert "\tTime Solve: 12.500000ms\n\t#iters solve: 3\n\n\nLocal Timings Per Frame:\n[ frame 0 ]\n" in log
    assert "\tTime Process Input: 0.500000ms\n\tTime Re-Integrate: 1.000000ms\n\tTime Reconstruct: 0.750000\n\tTime Visualize: 0.000000\n" in log
    assert log.endswith("Total Timings Per Frame:\n[ frame 0 ] 4 ms\n[ frame 1 ] 5 ms\n[ frame 2 ] 6 ms\n\n\n")
    loc = (out / "excel_local.txt").read_text().splitlines()
    assert loc[:8] == ["Average times:", "SIFT Detection,1.5,3", "SIFT Matching,0.5,3", "Corr Filter,0.125,3", "Misc,0,3", "Solve,2,3", "Re-Integrate,1,3", "Misc,1.25,3"]
    assert "Match Filter Key Point,0,0.125,0.25" in loc and "Process Input,0.25,0.5,0.75" in loc
    assert (out / "excel_total.txt").read_text() == "Per Frame Timings,4,5,6"
    assert (out / "excel_global.txt").read_text().splitlines()[5] == "Solve,12.5,1"


_RECORD_CPP = r'''
#include <limits>
#include "bundlefusion/bundlefusion.hpp"
using namespace bundlefusion;
struct Fake : RGBDSensor {
    std::vector<float> depth; std::vector<unsigned char> color; int k = 0;
    Fake() {
        std::memset(&m_desc, 0, sizeof m_desc);
        m_desc.depthWidth = m_desc.colorWidth = 8; m_desc.depthHeight = m_desc.colorHeight = 6;
        const mat4f I = mat4f::identity(); mat4f K = I; K(0, 0) = 500; K(1, 1) = 501; K(0, 2) = 3.5f; K(1, 2) = 2.5f;
        std::memcpy(m_desc.depthIntrinsics, K.m, 64); std::memcpy(m_desc.colorIntrinsics, K.m, 64);
        std::memcpy(m_desc.depthExtrinsics, I.m, 64); std::memcpy(m_desc.colorExtrinsics, I.m, 64);
        depth.assign(48, 0.0f); color.assign(48 * 4, 0);
    }
    bool processDepth() override {
        for (int i = 0; i < 48; ++i) { depth[i] = 1.0f + 0.0005f * (i + 100 * k); color[4 * i] = (unsigned char)(i + k); color[4 * i + 1] = (unsigned char)(2 * i); color[4 * i + 2] = 7; color[4 * i + 3] = 255; }
        depth[0] = -std::numeric_limits<float>::infinity(); depth[1] = 1.2345f; depth[2] = 0.0f;
        ++k; return true;
    }
    bool processColor() override { return true; }
    const float* getDepthFloat() const override { return depth.data(); }
    const unsigned char* getColorRGBX() const override { return color.data(); }
    std::string getSensorName() const override { return "FakeSensor"; }
};
int main(int argc, char** argv) {
    const std::string dir = argv[1];
    Fake s; s.setRecordTempPrefix(dir + "/tmp_");
    for (int i = 0; i < 3; ++i) { s.processDepth(); s.recordFrame(); }
    std::vector<mat4f> traj(2, mat4f::identity()); traj[1](0, 3) = 0.25f; traj[1](2, 3) = -1.5f;
    const std::string a = s.saveRecordedFramesToFile(dir + "/rec.sens", traj);            // rec.sens exists already -> rec1.sens
    std::printf("%s\n", a.c_str());
    for (int i = 0; i < 1; ++i) { s.processDepth(); s.recordFrame(); }
    std::vector<mat4f> three(3, mat4f::identity());
    try { s.saveRecordedFramesToFile(dir + "/x.sens", three, true); return 2; }           // more transforms than frames
    catch (const std::exception& e) { std::printf("%s\n", e.what()); }
    return 0;
}
'''


def test_rgbd_sensor_recording_writes_a_sens_file(built, tmp_path):
    """RGBDSensor::recordFrame / saveRecordedFramesToFile (RGBDSensor.cpp:264-312, 353-398) in bundlefusion.hpp."""
    from bundlefusion_amd import sensordata as sdm
    src = tmp_path / "rec.cpp"
    src.write_text(_RECORD_CPP)
    exe = tmp_path / "rec"
    libdir = os.path.join(ROOT, "bundlefusion_amd", "lib")
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), str(src), "-L", libdir, "-lbf_hip", "-Wl,-rpath," + libdir,
                        "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    (tmp_path / "rec.sens").write_bytes(b"already here")
    out = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.splitlines()
    assert lines[0] == str(tmp_path / "rec1.sens") and "more transforms than frames" in lines[1]
    assert (tmp_path / "rec.sens").read_bytes() == b"already here"                       # not overwritten
    assert not [p for p in os.listdir(tmp_path) if p.startswith("tmp_")]                  # the temporary recording is gone
    sd = sdm.SensorData(tmp_path / "rec1.sens")
    assert len(sd) == 2 and sd.sensor_name == "FakeSensor"                                # the third frame had no pose: dropped
    assert sd.info.depthShift == 1000.0 and sd.info.depthCompressionType == sdm.DEPTH_ZLIB_USHORT and sd.info.colorCompressionType == sdm.COLOR_JPEG
    assert np.array(sd.info.depthIntrinsic, np.float32).reshape(4, 4)[0, 0] == 500
    T1 = sd.pose(1)[0]
    assert T1[0, 3] == 0.25 and T1[2, 3] == -1.5 and np.array_equal(sd.pose(0)[0], np.eye(4, dtype=np.float32))
    for k in range(2):
        d = sd.depth_raw(k).reshape(-1)
        want = np.array([np.float32(1.0) + np.float32(0.0005) * np.float32(i + 100 * k) for i in range(48)], np.float32)
        q = np.floor(want * np.float32(1000.0) + np.float32(0.5)).astype(np.uint16)
        q[0] = 0; q[1] = 1235; q[2] = 0                                                    # -inf and 0 are invalid; round(1234.5) = 1235
        assert np.array_equal(d, q), k
        c = sd.color_rgbx(k).reshape(-1, 4).astype(int)                                        # JPEG, quality 90: close, not equal
        want_c = np.stack([np.arange(48) + k, 2 * np.arange(48), np.full(48, 7)], 1)
        assert np.abs(c[:, :3] - want_c).max() <= 6 and (c[:, 3] == 255).all()
    sd.close()


def test_committed_bench_line_follows_the_contract():
    """profiles/r01_bench.json is what `python bench.py` printed on the MI355X box: one JSON object with the driver's keys plus the
    roofline and cpu_baseline objects."""
    import json
    line = json.load(open(os.path.join(ROOT, "profiles", "r01_bench.json")))
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                     ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(line[key], typ), key
    assert "vs_baseline" in line and line["vs_baseline"] is None          # BASELINE.md holds no published number for this metric
    assert line["unit"] == "frames/s" and line["scaling"] in ("weak", "strong") and "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] - line["n_gpus"] * 1e3 / line["ms_per_step"]) < 1e-6 * line["value"]
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma", "valu") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None or r["traffic"] > 0
    c = line["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == line["unit"] and c["sample"]


def test_round6_bench_line_and_pmc_file_are_consistent():
    """profiles/r06_bench_driver.json (the driver's invocation on the MI355X box) against the contract and against the PMC file its `traffic` and `valu` figures come
    from: the counters were collected on the update kernels' CURRENT source AND build flags (bench.py reports neither otherwise), with the calibrated factor on record."""
    import json
    import sys
    line = json.loads(open(os.path.join(ROOT, "profiles", "r06_bench_driver.json")).read().strip().split("\n")[-1])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                     ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(line[key], typ), key
    assert line["steps"] == 20 and line["warmup"] == 5 and line["n_gpus"] == 1 and line["vs_baseline"] is None
    assert abs(line["value"] - 1e3 / line["ms_per_step"]) < 1e-6 * line["value"]
    assert line["config"]["solve_lag"] == 10                    # the library's default schedule (the reference's optimiser thread, deterministic)
    r = line["roofline"]
    assert r["bound"] == "valu" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["peak"] == 8000.0
    assert 0.0 < r["frac"] < 1.0 and r["frac_per_operator"] > r["frac"] and 0.0 < r["frac_beyond_l3"] < 1.0
    assert r["traffic"] is not None and r["traffic"] > r["algorithmic_bytes_per_launch"] * 0.5
    v = r["valu"]
    assert v is not None and 0.2 < v["frac_at_2_cycles_per_instruction"] < v["frac"] < 1.0 and 3.0 < v["cycles_per_instruction_measured"] < 6.0
    assert line["serial_order"]["solve_lag_frames"] == 0 and 0 < line["serial_order"]["value"] < 1.2 * line["value"]
    cs = line["class_surface"]
    assert cs["deferred_batching"]["value"] > cs["as_the_reference_issues_them"]["value"] > 0 and cs["deferred_batching"]["integrate"] == cs["as_the_reference_issues_them"]["integrate"]
    ls = line["long_stream"]
    assert ls["frames"] == 5000 and ls["frames_tracked"] == 5000 and ls["ate_optimized_m"] < 0.02 and len(ls["per_500_frames"]) == 10
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == line["unit"]
    sys.path.insert(0, ROOT)
    from tools.pmc_to_json import update_kernel_sha, build_flags_sha
    pmc = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_tsdf_update.json")))["fast"]
    assert pmc["update_kernel_sha256"] == update_kernel_sha(), "the committed PMC figures were collected on another version of the voxel-update kernels"
    assert pmc["build_flags_sha256"] == build_flags_sha(), "the committed PMC figures were collected on a build with other compiler flags"
    assert abs(pmc["fetch_factor_applied"] - pmc["calibration"]["k_probe_slices"]["fetch_factor"]) < 1e-12 and 2.0 < pmc["fetch_factor_applied"] < 3.0
    assert pmc["fused"]["launches"] > 200 and 12320 < pmc["fused"]["hbm_bytes_per_visited_block"] < 60000
    assert pmc["sq"]["launches"] > 200 and 2000 < pmc["sq"]["valu_wave_instructions_per_visited_block"] < 20000
    # the vector-memory return path (TD / TCP / VMEM passes of the same file) and its figure in the bench line
    mp = pmc["mem_pipe"]
    assert mp["compute_units"] == 224 and 0.3 < mp["td_busy_share_of_cu_cycles"] <= 1.0 and mp["td_stalled_on_l1_share_of_cu_cycles"] < mp["td_busy_share_of_cu_cycles"]
    assert 4.0 < mp["l1_accesses_per_vmem_instruction"] < 64.0 and 50 < mp["vmem_wave_instructions_per_visited_block"] < 500
    m = r["mem_pipe"]
    assert m is not None and 0.2 < m["td_busy_frac"] <= 1.0 and abs(m["td_busy_frac_in_the_pmc_run"] - mp["td_busy_share_of_cu_cycles"]) < 1e-9
    assert 0.3 < v["frac_in_the_pmc_run"] <= 1.05 and v["frac_in_the_pmc_run"] > v["frac_at_2_cycles_per_instruction"]


def test_bench_launches_its_own_ranks_and_refuses_a_mismatched_world():
    """`python bench.py --gpus N` without a launcher around it starts N ranks itself (torch.distributed.run on 127.0.0.1) instead of measuring
    one GPU under the label N; with a launcher whose world differs from --gpus it prints no line.  (--launch-check: the rendezvous over gloo
    only, no GPU work.)"""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    assert bench.launch_plan(1, {}, []) is None and bench.launch_plan(4, {"WORLD_SIZE": "4"}, []) is None
    plan = bench.launch_plan(4, {}, ["--gpus", "4", "--steps", "20"])
    assert plan[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in plan and plan[plan.index("--nproc-per-node") + 1] == "4"
    assert plan[plan.index("--master-addr") + 1] == "127.0.0.1" and plan[-4:] == ["--gpus", "4", "--steps", "20"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"launch_check": True, "n_gpus": 2}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True, timeout=300,
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "{" not in r.stdout and "mismatched" in r.stderr
    # more ranks than devices: refused at once, before any rank is started (this box has no GPU: 0 < 2)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "{" not in r.stdout and "GPU(s) are visible" in r.stderr


_POSEHELPER_CPP = r'''
#include "bundlefusion/bundlefusion.hpp"
using namespace bundlefusion;
static mat4f rotZ(float a, float tx, float ty, float tz) { mat4f T = mat4f::identity(); T(0,0)=std::cos(a); T(0,1)=-std::sin(a); T(1,0)=std::sin(a); T(1,1)=std::cos(a); T(0,3)=tx; T(1,3)=ty; T(2,3)=tz; return T; }
int main(int, char** argv) {
    std::vector<mat4f> ref, traj;
    const mat4f G = rotZ(0.7f, 1.0f, -2.0f, 0.5f);
    for (int i = 0; i < 12; ++i) { ref.push_back(rotZ(0.1f * i, 0.3f * i, 0.1f * i * i, 0.05f * i)); traj.push_back(G * ref.back()); }
    mat4f bad; bad.setZero(-std::numeric_limits<float>::infinity()); traj[4] = bad;
    if (PoseHelper::countNumValidTransforms(traj) != 11) return 1;
    const auto ate = PoseHelper::evaluateAteRmse(traj, ref);
    if (!(ate.first < 1e-5f) || ate.second != 11) return 2;
    const mat4f A = PoseHelper::getAlignmentBetweenTrajectories(traj, ref);      // maps trajectory positions onto the reference: inverse of G
    const mat4f I = A * G;
    for (int i = 0; i < 16; ++i) if (std::fabs(I.m[i] - ((i % 5 == 0) ? 1.0f : 0.0f)) > 1e-4f) return 3;
    const auto per = PoseHelper::evaluateErr2PerImage(traj, ref);
    if (per.size() != 11 || per[4].first != 5 || per[10].second > 1e-9f) return 4;
    std::vector<mat4f> all; for (int i = 0; i < 6; ++i) all.push_back(rotZ(0.0f, (float)i, 0, 0));
    std::vector<mat4f> keys = {rotZ(0.0f, 10.0f, 0, 0), rotZ(0.0f, 20.0f, 0, 0)};
    PoseHelper::composeTrajectory(3, keys, all);                                  // two chunks of three frames, re-based on their key poses
    const float want[6] = {10, 11, 12, 20, 21, 22};
    for (int i = 0; i < 6; ++i) if (std::fabs(all[i](0, 3) - want[i]) > 1e-5f) return 5;
    PoseHelper::saveToPoseFile(std::string(argv[1]) + "/poses.txt", traj);
    return 0;
}
'''


def test_pose_helper_functions(built, tmp_path):
    """PoseHelper (PoseHelper.h:8-166) in bundlefusion.hpp: valid-pose count, composeTrajectory, ATE, alignment, per-image error, pose file."""
    src = tmp_path / "ph.cpp"
    src.write_text(_POSEHELPER_CPP)
    exe = tmp_path / "ph"
    libdir = os.path.join(ROOT, "bundlefusion_amd", "lib")
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), str(src), "-L", libdir, "-lbf_hip", "-Wl,-rpath," + libdir,
                        "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.returncode
    rows = [l.split() for l in (tmp_path / "poses.txt").read_text().splitlines()]
    assert [int(r[0]) for r in rows] == [0, 1, 2, 3, 5, 6, 7, 8, 9, 10, 11]                   # the invalid pose is skipped, indices are kept
    q = np.array([[float(v) for v in r[4:]] for r in rows])
    assert np.allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-5) and np.allclose(q[:, :2], 0.0, atol=1e-6)     # rotations about z: (0, 0, sin, cos)
    ang = 2 * np.arctan2(q[:, 2], q[:, 3])
    assert np.allclose(ang, 0.7 + 0.1 * np.array([0, 1, 2, 3, 5, 6, 7, 8, 9, 10, 11]), atol=1e-4)
    assert abs(float(rows[0][1]) - 1.0) < 1e-6 and abs(float(rows[0][2]) + 2.0) < 1e-6 and abs(float(rows[0][3]) - 0.5) < 1e-6


def test_chunked_runner_local_half_runs_ahead_and_fails_loudly():
    """ChunkedRunner (world 1): the local half of the next chunk is produced on a second thread while the current one is consumed - same
    schedule with and without the run-ahead; an error inside the local half surfaces on the main thread, never as a silent empty package."""
    import numpy as np
    import pytest
    from bundlefusion_amd.shard import ChunkedRunner
    S = 10

    class Worker:
        package_bytes = 16
        def __init__(self, fail_at=None): self.ran, self.fail_at = [], fail_at
        def run(self, chunk, frames, out=None):
            if chunk == self.fail_at:
                raise RuntimeError("local half failed for chunk %d" % chunk)
            out[:4].view(np.int32)[0] = chunk
            self.ran.append(chunk)

    class Pipe:
        def __init__(self): self.log = []
        def process_frame_chunked(self, d, c, pkg, j):
            self.log.append((d, int(pkg[:4].view(np.int32)[0]), j)); return True

    feed = [(i, i) for i in range(1 + 4 * S)]
    logs = []
    for prefetch in (True, False):
        r = ChunkedRunner(Pipe(), Worker(), feed, S, prefetch=prefetch)
        r.advance(13); r.wait(); r.advance(len(feed) - 13); r.close()
        assert r.worker.ran == [0, 1, 2, 3] and r.local_chunks == 4 and r.rounds == 4
        logs.append(r.pipe.log)
    assert logs[0] == logs[1] and [f for f, _, _ in logs[0]] == list(range(len(feed)))
    assert all(c == (0 if f == 0 else (f - 1) // S) for f, c, _ in logs[0])
    r = ChunkedRunner(Pipe(), Worker(fail_at=2), feed, S)
    with pytest.raises(RuntimeError, match="chunk 2"):
        r.advance(len(feed))
    assert max(f for f, _, _ in r.pipe.log) <= 2 * S         # nothing of chunk 2 was consumed


def test_timed_window_of_the_chunk_parallel_mode_contains_chunk_local_work():
    """bench.py's window over the chunk-parallel mode (shard.timed_window): it starts at a round boundary, and the stream is long enough for the
    local halves of the NEXT round to run inside it - a window over the replicated global half alone would not be a whole-loop measurement."""
    import numpy as np
    from bundlefusion_amd.shard import ChunkedRunner, timed_window, chunk_frames
    S = 10
    for world in (1, 2, 4, 8):
        for steps in (20, 200):
            pre, total, n = timed_window(205, steps, world, S)
            rnd = world * S
            assert pre >= 205 and (pre - 1) % rnd == 0 and total == pre + steps
            first_round = (pre - 1) // S                               # first chunk of the round the window starts with
            last_round = ((total - 2) // S) // world * world           # ... and of the round it ends in
            for rank in range(world):                                  # every rank's chunk of the round behind the last one is in the stream
                assert chunk_frames(last_round + world + rank, S)[1] < n
            assert first_round % world == 0

    class Worker:
        package_bytes = 16
        def __init__(self): self.ran = []
        def run(self, chunk, frames, out=None):
            out[:4].view(np.int32)[0] = chunk; self.ran.append(chunk)

    class Pipe:
        def process_frame_chunked(self, d, c, pkg, j): return True

    pre, total, n = timed_window(205, 20, 1, S)
    r = ChunkedRunner(Pipe(), Worker(), [(i, i) for i in range(n)], S)
    r.advance(pre); r.wait()
    runs0, rounds0 = r.local_runs, r.rounds
    r.advance(20); r.wait()
    assert r.local_runs - runs0 == 2 and r.rounds - rounds0 == 2       # world 1, 20 frames = two rounds: two all-gathers, two local halves - the steady state
    r.close()


def test_processed_summary_file(built, tmp_path):
    """processed.txt as StopScanningAndExit writes it (DepthSensing.cpp:921-957): the validity rule and the four lines."""
    from bundlefusion_amd.capi import write_processed_summary
    T = np.tile(np.eye(4, dtype=np.float32), (7, 1, 1))
    T[4:] = -np.inf                                                      # 4 of 7 valid: 4 >= round(3.5) = 4
    p = tmp_path / "processed.txt"
    assert write_processed_summary(p, 12345, T) is True
    assert p.read_text() == "valid = true\nheapFreeCount = 12345\nnumValidOptTransforms = 4\nnumTransforms = 7\n"
    T[3] = -np.inf                                                       # 3 of 7: not enough valid transforms
    assert write_processed_summary(p, 12345, T) is False and p.read_text().startswith("valid = false\nheapFreeCount = 12345\nnumValidOptTransforms = 3\n")
    T[3] = np.eye(4)
    assert write_processed_summary(p, 799, T) is False                   # the heap is (almost) used up
    assert write_processed_summary(p, 800, T) is True
    assert write_processed_summary(p, 5000, T, aborted=True) is False and p.read_text() == "valid = false\nABORTED\n"


def test_comm_callback_transport_and_single_rank_exchange(built):
    """include/bf_comm.h without a device: a callback communicator hands the caller's pointers to the host's all-gather and reports its failure; a world of one
    exchanges chunk packages without touching the transport; bad arguments are refused."""
    from bundlefusion_amd import capi
    from bundlefusion_amd.capi import lib
    calls = []

    def gather(user, send, recv, nbytes, stream):          # "rank 1 of 3": the host's own transport; here it fills all three slots from the one buffer it has
        calls.append((int(nbytes), stream))
        if nbytes == 13:
            return 7
        for r in range(3):
            C.memmove(recv + r * nbytes, send, nbytes)
            C.memset(recv + r * nbytes, r, 1)
        return 0
    cb = capi._ALL_GATHER_FN(gather)
    h = C.c_void_p()
    assert lib.bf_comm_create_callback(cb, None, 3, 3, C.byref(h)) != 0            # rank >= world
    assert lib.bf_comm_create_callback(cb, None, 3, 1, C.byref(h)) == 0
    w, r = C.c_uint32(), C.c_uint32()
    assert lib.bf_comm_world(h, C.byref(w), C.byref(r)) == 0 and (w.value, r.value) == (3, 1)
    send = np.arange(64, dtype=np.uint8); recv = np.zeros(3 * 64, np.uint8)
    assert lib.bf_comm_all_gather(h, send.ctypes.data_as(C.c_void_p), recv.ctypes.data_as(C.c_void_p), C.c_uint64(64), None) == 0
    assert calls == [(64, None)]
    for k in range(3):
        assert recv[k * 64] == k and np.array_equal(recv[k * 64 + 1:(k + 1) * 64], send[1:])
    assert lib.bf_comm_all_gather(h, send.ctypes.data_as(C.c_void_p), recv.ctypes.data_as(C.c_void_p), C.c_uint64(0), None) == 0 and len(calls) == 1      # nothing to do
    assert lib.bf_comm_all_gather(h, send.ctypes.data_as(C.c_void_p), recv.ctypes.data_as(C.c_void_p), C.c_uint64(13), None) != 0                           # the host's failure is reported
    assert b"callback failed with 7" in lib.bf_last_error()
    assert lib.bf_comm_all_gather(h, None, recv.ctypes.data_as(C.c_void_p), C.c_uint64(8), None) != 0
    assert lib.bf_comm_destroy(h) == 0
    # a world of one: bf_chunk_exchange is a copy, the transport is never called
    calls.clear()
    h1 = C.c_void_p()
    assert lib.bf_comm_create_callback(cb, None, 1, 0, C.byref(h1)) == 0
    mine = np.random.RandomState(3).randint(0, 256, 1000).astype(np.uint8); got = np.zeros_like(mine)
    assert lib.bf_chunk_exchange(h1, mine.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p), C.c_uint64(mine.size), None) == 0
    assert np.array_equal(got, mine) and not calls
    assert lib.bf_chunk_exchange(h1, mine.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p), C.c_uint64(0), None) != 0
    assert lib.bf_comm_destroy(h1) == 0 and lib.bf_comm_destroy(None) == 0


def test_sweep_key_capacity_covers_the_measured_key_counts():
    """tools/tsdf_sweep.py --comm-alloc (bench.py's `sweep` block at N > 1): the per-rank key capacity must cover the keys a rank's band collects - measured ~210 k distinct
    in-frustum blocks per 1280x960 frame at 2 mm and ~15 k per 640x480 frame at 4 mm - with a factor of two, at every rank count the driver uses."""
    from tools.tsdf_sweep import alloc_comm_capacity
    for world in (2, 4, 8):
        assert alloc_comm_capacity(1280, 960, 0.002, world) >= 2 * 210000 / world
        assert alloc_comm_capacity(640, 480, 0.004, world) >= 2 * 15000 / world
        cap = alloc_comm_capacity(1280, 960, 0.002, world)
        assert cap & (cap - 1) == 0 and cap * 8 * world <= 64 << 20          # a power of two; the gathered records of one operator stay below 64 MB
