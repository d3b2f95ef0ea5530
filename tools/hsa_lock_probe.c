/* How does this ROCm stack page-lock ordinary host memory (hipHostRegister -> hsa_amd_memory_lock), and what happens
 * to a lock when a NEIGHBOURING range that shares a 4 KiB page with it is locked and unlocked?  No GPU access is made:
 * the state is read back with hsa_amd_pointer_info and hsa_amd_svm_attributes_get(ACCESS_QUERY).
 *   gcc -O1 -I/opt/rocm/include tools/hsa_lock_probe.c -L/opt/rocm/lib -lhsa-runtime64 -o /tmp/hsa_lock_probe && /tmp/hsa_lock_probe */
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

static hsa_agent_t gpu;
static int have_gpu = 0;
static hsa_status_t pick(hsa_agent_t a, void* d) {
    hsa_device_type_t t;
    (void)d;
    hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !have_gpu) gpu = a, have_gpu = 1;
    return HSA_STATUS_SUCCESS;
}

static void show(const char* what, void* p) {
    hsa_amd_pointer_info_t info;
    memset(&info, 0, sizeof(info));
    info.size = sizeof(info);
    hsa_status_t s = hsa_amd_pointer_info(p, &info, NULL, NULL, NULL);
    hsa_amd_svm_attribute_pair_t q = {HSA_AMD_SVM_ATTRIB_ACCESS_QUERY, gpu.handle};
    hsa_status_t s2 = hsa_amd_svm_attributes_get((void*)((uintptr_t)p & ~(uintptr_t)4095), 4096, &q, 1);
    printf("  %-34s %p: pointer_info st=%d type=%d agentBase=%p hostBase=%p size=%zu | svm query st=%#x attr=%#llx\n", what, p, (int)s,
           (int)info.type, info.agentBaseAddress, info.hostBaseAddress, info.sizeInBytes, (unsigned)s2, (unsigned long long)q.attribute);
}

int main(void) {
    if (hsa_init() != HSA_STATUS_SUCCESS) return 1;
    hsa_iterate_agents(pick, NULL);
    if (!have_gpu) return 2;
    char* base = (char*)malloc(1 << 16);
    memset(base, 1, 1 << 16);
    char* page = (char*)(((uintptr_t)base + 4095) & ~(uintptr_t)4095);
    char *A = page + 128, *Aend = page + 2 * 4096 + 2000; /* A ends inside page 2 */
    char *B = Aend, *Bend = page + 5 * 4096 + 100;          /* B starts in the same page */
    printf("heap block %p; A=[%p,%p) B=[%p,%p); shared page %p; type codes: 0 unknown 1 hsa 2 locked 3 graphics 4 ipc\n", (void*)base,
           (void*)A, (void*)Aend, (void*)B, (void*)Bend, (void*)(page + 2 * 4096));
    show("A page0, before", A);
    void *da = NULL, *db = NULL;
    hsa_status_t s = hsa_amd_memory_lock(A, (size_t)(Aend - A), &gpu, 1, &da);
    printf("lock A -> st=%#x agent_ptr=%p (same as host: %d)\n", (unsigned)s, da, da == (void*)A);
    show("A page0, A locked", A);
    show("shared page (A's last), A locked", Aend - 8);
    s = hsa_amd_memory_lock(B, (size_t)(Bend - B), &gpu, 1, &db);
    printf("lock B (shares a page with A) -> st=%#x agent_ptr=%p\n", (unsigned)s, db);
    show("shared page, A and B locked", Aend - 8);
    show("B interior", B + 4096);
    if (s == HSA_STATUS_SUCCESS) {
        s = hsa_amd_memory_unlock(B);
        printf("unlock B -> st=%#x\n", (unsigned)s);
    }
    show("shared page (A's last), B unlocked", Aend - 8);
    show("A page0, B unlocked", A);
    s = hsa_amd_memory_unlock(A);
    printf("unlock A -> st=%#x\n", (unsigned)s);
    show("A page0, A unlocked", A);
    /* What is left of a lock when the memory is unmapped and NEW memory appears at the same address (free + malloc of a
     * large block, or a heap that shrinks and grows again)?  A cache of locked ranges keyed by address would call it a hit. */
    size_t len = 8 << 20;
    char* m = (char*)mmap(NULL, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    memset(m, 1, len);
    void* dm = NULL;
    s = hsa_amd_memory_lock(m, len, &gpu, 1, &dm);
    printf("mmap block %p locked -> st=%#x\n", (void*)m, (unsigned)s);
    show("mmap block, locked", m + 4096);
    hsa_amd_memory_unlock(m);
    show("mmap block, unlocked", m + 4096);
    munmap(m, len);
    char* m2 = (char*)mmap(m, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_FIXED, -1, 0);
    memset(m2, 2, len);
    printf("munmap + new mmap at the same address %p (same: %d)\n", (void*)m2, m2 == m);
    show("NEW memory at the old address", m2 + 4096);
    hsa_shut_down();
    return 0;
}
