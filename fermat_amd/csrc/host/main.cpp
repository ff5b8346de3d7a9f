// main.cpp — `fermat_hip`, the batch renderer (src/main.cu:98-218) on top of the host mirror.
//
//   fermat_hip -i scene.{fa,obj} [-r W H] [-a aspect] [-c camera.txt] [-pt | -bpt | -psfpt] [-passes N] [-o output] [-ref ref.tga]
//              [-benchmark file] [-save-intermediate] [PT flags: -pl/-bounces/-nee/-bsdf/-nee-alg mesh|vpl ...]
//              [-data dir] [-device id] [-filtered | -shading-mode N]   (kFiltered = EAW-denoised output; the reference toggles it in the viewer)
//              [-gpus N]   one process per GPU of this node (forked here), image rows interleaved over the ranks, frame gathered to rank 0
//                          over RCCL (fpt_gather_framebuffer); -pt and -bpt
//   fermat_hip -diff a.tga b.tga
// As in the reference the pass loop runs i = 0..N inclusive (N+1 samples per pixel), the image is written as <output>.tga
// through to_rgba, and -ref prints the RMSE of the 8-bit image against a reference TGA (diff_image, src/main.cu:63-96).
#include "scene_io.h"
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <sys/wait.h>

using namespace fermat;

namespace {

bool load_image(const char* filename, std::vector<float>& img, int& w, int& h)
{
	int bits = 0;
	unsigned char* rgb = load_tga(filename, &w, &h, &bits);
	if (!rgb) return false;
	const size_t nb = size_t(bits) >> 3;
	img.resize(size_t(w) * h * 3);
	for (size_t p = 0; p < size_t(w) * h; ++p) for (int c = 0; c < 3; ++c) img[p * 3 + c] = float(rgb[p * nb + c]) / 255.0f;
	delete[] rgb;
	return true;
}

unsigned char quantize255(float x) { const float v = x * 255.0f; return (unsigned char)(v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v)); }

// RMSE = sqrt(mean_p |dst-ref|^2) and a false-colour error image (blue -> yellow below 0.5, yellow -> red above)
float diff_image(int w, int h, const float* ref, float* dst, unsigned char* rgb)
{
	const size_t n = size_t(w) * h;
	float e_sum = 0.0f;
	for (size_t p = 0; p < n; ++p)
	{
		float e2 = 0.0f;
		for (int c = 0; c < 3; ++c) { dst[p * 3 + c] -= ref[p * 3 + c]; e2 += dst[p * 3 + c] * dst[p * 3 + c]; }
		e_sum += std::fabs(e2) / float(n);
	}
	e_sum = std::sqrt(e_sum);
	const float col1[3] = { 0.2f, 0.3f, 0.9f }, col2[3] = { 1.0f, 0.9f, 0.2f }, col3[3] = { 1.0f, 0.0f, 0.0f };
	for (size_t p = 0; p < n; ++p)
	{
		const float e = std::max(std::fabs(dst[p * 3]), std::max(std::fabs(dst[p * 3 + 1]), std::fabs(dst[p * 3 + 2])));
		for (int c = 0; c < 3; ++c)
		{
			const float t = e < 0.5f ? e / 0.5f : std::sqrt(e - 0.5f) / 0.5f;
			const float v = e < 0.5f ? col1[c] * (1.0f - t) + col2[c] * t : col2[c] * (1.0f - t) + col3[c] * t;
			rgb[p * 3 + c] = quantize255(v);
		}
	}
	return e_sum;
}

std::string default_data_dir()
{
	if (FILE* f = std::fopen("glossy_reflectance.dat", "rb")) { std::fclose(f); return "."; }      // the reference's lookup: the CWD
	char exe[4096]; const ssize_t n = readlink("/proc/self/exe", exe, sizeof(exe) - 1);
	if (n > 0) { exe[n] = '\0'; std::string p(exe); const size_t k = p.find_last_of('/'); if (k != std::string::npos) return p.substr(0, k) + "/../data"; }
	return ".";
}

} // namespace

int main(int argc, char** argv)
{
	if (argc > 3 && std::strcmp(argv[1], "-diff") == 0)
	{
		std::vector<float> a, b; int w1, h1, w2, h2;
		if (!load_image(argv[2], a, w1, h1) || !load_image(argv[3], b, w2, h2)) { std::fprintf(stderr, "error: cannot load the images\n"); return 1; }
		if (w1 != w2 || h1 != h2) { std::fprintf(stderr, "error: differing image resolutions!\n"); return 1; }
		std::vector<unsigned char> rgb(size_t(w1) * h1 * 3);
		std::fprintf(stderr, "RMSE: %f\n", diff_image(w1, h1, a.data(), b.data(), rgb.data()));
		write_tga("diff.tga", w1, h1, rgb.data(), 3);
		return 0;
	}

	const char* filename = nullptr; const char* output_name = "output"; const char* camera_file = nullptr; const char* bench_name = nullptr;
	std::string data_dir = default_data_dir();
	uint32 n_passes = 1024;
	int n_gpus = 1;
	bool save_intermediate = false;
	std::vector<float> ref_img; int ref_w = 0, ref_h = 0;
	for (int i = 1; i < argc; ++i)
	{
		auto is = [&](const char* f) { return std::strcmp(argv[i], f) == 0; };
		if (is("-i") && i + 1 < argc) filename = argv[++i];
		else if (is("-o") && i + 1 < argc) output_name = argv[++i];
		else if (is("-c") && i + 1 < argc) camera_file = argv[++i];
		else if (is("-passes") && i + 1 < argc) n_passes = uint32(std::atoi(argv[++i]));
		else if (is("-gpus") && i + 1 < argc) n_gpus = std::max(1, std::atoi(argv[++i]));
		else if (is("-benchmark") && i + 1 < argc) bench_name = argv[++i];
		else if (is("-data") && i + 1 < argc) data_dir = argv[++i];
		else if (is("-save-intermediate")) save_intermediate = true;
		else if (is("-ref") && i + 1 < argc)
		{
			std::fprintf(stderr, "loading reference image... started (%s)\n", argv[i + 1]);
			if (!load_image(argv[++i], ref_img, ref_w, ref_h)) std::fprintf(stderr, "warning: failed to load %s\n", argv[i]);
			std::fprintf(stderr, "loading reference image... done (%d, %d)\n", ref_w, ref_h);
		}
		else if (is("-view")) { std::fprintf(stderr, "the interactive viewer is not part of this build\n"); return 1; }
	}
	if (!filename)
	{
		std::fprintf(stderr, "options:\n  -i scene.obj|scene.fa  specify the input scene\n  -r int int             specify the resolution\n"
		                     "  -a float               specify the aspect ratio\n  -c camera.txt          specify a camera file\n"
		                     "  -pt                    use the PT renderer\n  -passes int            number of passes - 1\n  -o name                output image name\n");
		return 0;
	}
	// -gpus N: fork the other ranks BEFORE anything touches HIP or RCCL; rank 0 creates the RCCL id and hands it down one pipe per child
	int rank = 0;
	char comm_id[FPT_COMM_ID_BYTES]; std::memset(comm_id, 0, sizeof(comm_id));
	std::vector<pid_t> children;
	if (n_gpus > 1)
	{
		std::vector<int> write_ends;
		for (int r = 1; r < n_gpus; ++r)
		{
			int fd[2];
			if (pipe(fd) != 0) { std::perror("pipe"); return 1; }
			const pid_t pid = fork();
			if (pid < 0) { std::perror("fork"); return 1; }
			if (pid == 0)
			{
				close(fd[1]); for (int w : write_ends) close(w);
				rank = r;
				size_t got = 0;
				while (got < sizeof(comm_id)) { const ssize_t k = read(fd[0], comm_id + got, sizeof(comm_id) - got); if (k <= 0) { std::fprintf(stderr, "rank %d: no RCCL id from rank 0\n", r); return 1; } got += size_t(k); }
				close(fd[0]);
				children.clear();
				break;
			}
			close(fd[0]); write_ends.push_back(fd[1]); children.push_back(pid);
		}
		if (rank == 0)
		{
			if (fpt_comm_unique_id(comm_id) != 0) { std::fprintf(stderr, "error: %s\n", fpt_comm_last_error()); return 1; }
			for (int w : write_ends) { if (write(w, comm_id, sizeof(comm_id)) != ssize_t(sizeof(comm_id))) { std::perror("write"); return 1; } close(w); }
		}
	}
	int status = 0;
	try
	{
		if (rank == 0) std::fprintf(stderr, "loading mesh file %s... started\n", filename);
		HostScene scene;
		scene.load(filename, data_dir.c_str());
		std::fprintf(stderr, "  bbox[%f, %f, %f][%f, %f, %f]\n", scene.bbox[0], scene.bbox[1], scene.bbox[2], scene.bbox[3], scene.bbox[4], scene.bbox[5]);
		std::fprintf(stderr, "loading mesh file %s... done\n  triangles : %d\n  vertices  : %d\n  materials : %d\n  groups    : %d\n  textures  : %d\n",
		             filename, scene.mesh.num_triangles, scene.mesh.num_vertices, int(scene.mesh.materials.size()), int(scene.mesh.group_names.size()), int(scene.mesh.textures.size()));
		fpt_camera cam; bool override_camera = false;
		if (camera_file)
		{
			cam = scene.arrays(nullptr).camera;
			if (!load_camera_file(camera_file, cam)) { std::fprintf(stderr, "failed opening camera file %s\n", camera_file); return 1; }
			override_camera = true;
		}
		const SceneArrays arrays = scene.arrays(override_camera ? &cam : nullptr);
		RenderingContext renderer;
		if (n_gpus > 1) renderer.set_sharding(rank, n_gpus, comm_id);
		renderer.init(argc, argv, arrays);
		const uint32 W = renderer.res().x, H = renderer.res().y;
		std::vector<uint8_t> rgba(size_t(W) * H * 4);
		const auto t_render = std::chrono::steady_clock::now();
		for (uint32 i = 0; i <= n_passes; ++i)
		{
			renderer.render(i);
			if (i == n_passes || (save_intermediate && ((i + 1) & i) == 0))
			{
				renderer.gather_frame(0);                    // N > 1: every rank's rows travel to rank 0 (collective)
				if (rank != 0) continue;
				renderer.download_rgba(rgba.data());
				if (i == n_passes)
				{
					// (render() may only have recorded the passes: the download above is what makes the library finish them)
					const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_render).count();
					std::fprintf(stderr, "\n%u passes of %u x %u in %.3f s: %.1f Msample/s\n", n_passes + 1, W, H, sec, double(W) * H * (n_passes + 1) / sec * 1.0e-6);
				}
				char name[1024];
				if (save_intermediate) std::snprintf(name, sizeof(name), "%s-%u.tga", output_name, i + 1);
				else std::snprintf(name, sizeof(name), "%s.tga", output_name);
				std::fprintf(stderr, "\nsaving %s\n", name);
				write_tga(name, int(W), int(H), rgba.data(), 4);
				if (ref_w == int(W) && ref_h == int(H))
				{
					std::vector<float> img(size_t(W) * H * 3); std::vector<unsigned char> rgb(size_t(W) * H * 3);
					for (size_t p = 0; p < size_t(W) * H; ++p) for (int c = 0; c < 3; ++c) img[p * 3 + c] = float(rgba[p * 4 + c]) / 255.0f;
					std::fprintf(stderr, "RMSE: %f\n", diff_image(int(W), int(H), ref_img.data(), img.data(), rgb.data()));
					std::snprintf(name, sizeof(name), "%s-%u-diff.tga", output_name, i + 1);
					write_tga(name, int(W), int(H), rgb.data(), 3);
				}
			}
		}
		if (bench_name && rank == 0)
		{
			if (FILE* f = std::fopen(bench_name, "w")) { renderer.m_renderer->dump_speed_stats(f); std::fclose(f); }
			else std::fprintf(stderr, "warning: failed to open file %s\n", bench_name);
		}
	}
	catch (const std::exception& e) { std::fprintf(stderr, "error (rank %d): %s\n", rank, e.what()); status = 1; }
	for (pid_t pid : children) { int st = 0; waitpid(pid, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) status = 1; }
	return status;
}
