// hypre_app INPUT_FILE -- the mini-app's process bootstrap
// (/root/reference/src/main.cpp:31-229) for one process per MI355X.
//
// Launch: one process per GPU with RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR /
// MASTER_PORT in the environment (python -m torch.distributed.run --no-python
// ./hypre_app input.yaml does that), or a single process with nothing set.
#ifndef MI_HOST_WITH_LIBHYPRE
#include <hip/hip_runtime.h>
#endif

#include <chrono>
#include <cstdio>
#include <iostream>
#include <string>
#include <vector>

#include "HypreSystem.h"

// device of this rank: the node-local rank modulo the visible devices, chosen
// BEFORE HYPRE_Init (reference: getDevice(), src/main.cpp:9-29, :59-66)
static int pick_device(int count) {
  const int local = mi_env_int("LOCAL_RANK", mi_env_int("OMPI_COMM_WORLD_LOCAL_RANK", mi_env_int("RANK", 0)));
  return count > 0 ? local % count : 0;
}

int main(int argc, char *argv[]) {
  MPI_Init(&argc, &argv);
#ifdef MI_HOST_WITH_LIBHYPRE
  // opt-in adapter build against a real (CPU) libHYPRE: host memory, host execution, no device
  int iproc = 0, nproc = 1;
  MPI_Comm_rank(MPI_COMM_WORLD, &iproc);
  MPI_Comm_size(MPI_COMM_WORLD, &nproc);
  if (HYPRE_Init()) return 2;
  (void)pick_device;
#else
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
    std::cerr << "ERROR!! hypre_app needs a HIP device (MI355X); none is visible and there is no CPU path."
              << std::endl;
    return 2;
  }
  int device = pick_device(count);
  (void)hipSetDevice(device);
  HYPRE_Int ret = HYPRE_Init();
  if (ret) return 2;
  if (HYPRE_MI_CommInitFromEnv()) return 2;
  int iproc = 0, nproc = 1;
  MPI_Comm_rank(MPI_COMM_WORLD, &iproc);
  MPI_Comm_size(MPI_COMM_WORLD, &nproc);
  {
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, device);
    printf("rank=%d : %s : %s arch=%s : device=%d of %d : free memory=%1.8g GB, total memory=%1.8g GB\n", iproc,
           __FUNCTION__, prop.name, prop.gcnArchName, device, count, free_b / 1.e9, total_b / 1.e9);
  }
#endif
  fflush(stdout);
  MPI_Barrier(MPI_COMM_WORLD);
  auto start = std::chrono::steady_clock::now();

  if (argc != 2) {
    std::cout << "ERROR!! Incorrect arguments passed to program." << std::endl
              << "Usage: hypre_app INPUT_FILE" << std::endl
              << std::endl;
    return 1;
  }
  int rc = 0;
  try {
    YAML::Node inpfile = YAML::LoadFile(argv[1]);
    YAML::Node node = inpfile["solver_settings"];

    // memory / execution policy and the vendor-kernel knobs (src/main.cpp:97-156):
    // the library is device-only and has no vendor paths, the knobs are accepted
#ifdef MI_HOST_WITH_LIBHYPRE
    HYPRE_SetMemoryLocation(HYPRE_MEMORY_HOST);
    HYPRE_SetExecutionPolicy(HYPRE_EXEC_HOST);
#else
    HYPRE_SetGPUMemoryPoolSize(8, 3, 9, 2000LL * 1024 * 1024);
    HYPRE_SetUmpireDevicePoolName("HYPRE_DEVICE_POOL");
    HYPRE_SetUmpireDevicePoolSize((size_t)nalu::get_optional(node, "umpire_device_pool_mbs", 4096) * 1024 * 1024);
    HYPRE_SetMemoryLocation(HYPRE_MEMORY_DEVICE);
    HYPRE_SetExecutionPolicy(HYPRE_EXEC_DEVICE);
    HYPRE_SetSpGemmUseVendor(nalu::get_optional(node, "spgemm_use_vendor", 0) == 1);
    HYPRE_SetSpMVUseVendor(nalu::get_optional(node, "spmv_use_vendor", 0) == 1);
    HYPRE_SetSpTransUseVendor(nalu::get_optional(node, "sptrans_use_vendor", 0) == 1);
#endif

    const std::string csv_profile_file = nalu::get_optional<std::string>(node, "csv_profile_file", "");
    std::vector<std::string> names;
    std::vector<std::vector<double>> data;
    const int num_tests = nalu::get_optional(node, "num_tests", 1);
    for (int i = 0; i < num_tests; ++i) {
#ifndef MI_HOST_WITH_LIBHYPRE
      hypre_ResetDeviceRandGenerator(1234ULL, 0ULL);
#endif
      nalu::HypreSystem linsys(MPI_COMM_WORLD, inpfile);
      linsys.setup_precon_and_solver();
      linsys.checkMemory();
      linsys.load();
      linsys.checkMemory();
      linsys.solve();
      linsys.check_solution();
      linsys.output_linear_system();
      linsys.summarize_timers();
      if (!csv_profile_file.empty()) linsys.retrieve_timers(names, data);
      MPI_Barrier(MPI_COMM_WORLD);
      const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
      if (iproc == 0) std::cout << "Total time: " << elapsed << " seconds" << std::endl;
      if (!linsys.all_close()) rc = 3;
      linsys.destroy_system();
    }
    if (!csv_profile_file.empty() && iproc == 0 && !names.empty()) {
      FILE *fid = fopen(csv_profile_file.c_str(), "wt");
      if (fid) {
        for (size_t i = 0; i < names.size(); ++i) fprintf(fid, "%s%s", names[i].c_str(), i + 1 < names.size() ? "," : "\n");
        for (size_t j = 0; j < data[0].size(); ++j)
          for (size_t i = 0; i < names.size(); ++i)
            fprintf(fid, "%1.15g%s", j < data[i].size() ? data[i][j] : 0.0, i + 1 < names.size() ? "," : "\n");
        fclose(fid);
      }
    }
  } catch (const std::exception &e) {
    std::cerr << "rank " << iproc << " : ERROR : " << e.what() << std::endl;
    rc = 1;
  }
#ifdef MI_HOST_WITH_LIBHYPRE
  HYPRE_Finalize();
  MPI_Finalize();
#else
  MPI_Finalize();  // = HYPRE_MI_CommFinalize: the communicator lives in the library
  HYPRE_Finalize();
#endif
  (void)nproc;
  return rc;
}
