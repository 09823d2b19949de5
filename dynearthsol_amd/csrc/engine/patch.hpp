// engine/patch.hpp -- Node-block patches for the EN3 pass (passes/en3.hpp), built once per mesh.
// Part of the single translation unit des_dev.hip (included inside namespace des_hip, after
// DevClock / struct des_dev); not a stand-alone header.

// For every block of `npb` consecutive nodes (engine order):
//   pn_id  [pn_ptr[b] .. pn_ptr[b+1])   the nodes of its elements that are NOT its own, ascending
//   pe_*   [pe_ptr[b] .. pe_ptr[b+1])   its patch = every element touching one of its nodes, ascending:
//          pe_elem  element id | 1 << 30 if this block owns the element (holds its lowest node)
//          (ln)     local id of the element's four nodes in connectivity order: n - n0 for the
//                   block's own nodes, nown + position in pn_id for the others -- in pe_pack only
//          pe_slot  for each of the four nodes: position of that incidence in the block's slice of the
//                   CSR support list (k - sup_idx[n0]) if the node is the block's own, else -1
//   pe_pack  the three of them in ONE 16-byte record, the only form that goes to the device (one request per listed
//            element instead of three; EN2 reads the same records just ahead of EN3, which then finds them cached):
//            .x = elem (31 bits incl. the owner flag) | ln0 << 31 | ln1 << 40 | ln2 << 49
//            .y = ln3 | slot0 << 9 | slot1 << 21 | slot2 << 33 | slot3 << 45      (9-bit local ids, 12-bit slots, 0xfff = none)
struct PatchLists {
    int npb = 0, nb = 0, max_inc = 0, max_pn = 0, max_pe = 0;
    std::vector<int> pe_ptr, pe_elem, pn_ptr, pn_id;
    std::vector<ulonglong2> pe_pack;
};

// false: a block exceeds the LDS caps of the kernel (cap_inc incidences, cap_pn patch nodes)
bool build_patches(const des_mesh *m, int npb, int cap_inc, int cap_pn, PatchLists &P)
{
    const int nn = m->nnode, ne = m->nelem;
    const int *conn = m->connectivity, *sidx = m->support_idx, *sarr = m->support_arr, *slid = m->support_lidx;
    P = PatchLists();
    P.npb = npb; P.nb = (nn + npb - 1) / npb;
    P.pe_ptr.assign(1, 0); P.pn_ptr.assign(1, 0);
    std::vector<int> emark((size_t)ne, -1), nmark((size_t)nn, -1), elems, halo;
    std::vector<short> slots;                        // [4 * patch position]
    for (int b = 0; b < P.nb; ++b) {
        const int n0 = b * npb, n1 = std::min(nn, n0 + npb), nown = n1 - n0;
        const int kb = sidx[n0], ke = sidx[n1];
        if (ke - kb > cap_inc || ke - kb > 4094) return false;                               // (12-bit slots in pe_pack)
        elems.clear(); halo.clear();
        for (int k = kb; k < ke; ++k) if (emark[sarr[k]] != b) { emark[sarr[k]] = b; elems.push_back(sarr[k]); }
        std::sort(elems.begin(), elems.end());
        for (size_t q = 0; q < elems.size(); ++q) {
            const int e = elems[q];
            for (int i = 0; i < 4; ++i) {
                const int n = conn[(size_t)i*ne + e];
                if ((n < n0 || n >= n1) && nmark[n] != b) { nmark[n] = b; halo.push_back(n); }
            }
        }
        std::sort(halo.begin(), halo.end());
        if (nown + (int)halo.size() > cap_pn || nown + halo.size() > 511) return false;      // (9-bit local ids in pe_pack)
        // position of each element in the sorted patch, then the slots from the CSR slice
        for (size_t q = 0; q < elems.size(); ++q) emark[elems[q]] = -2 - (int)q;       // (restored to b below)
        slots.assign(4 * elems.size(), (short)-1);
        for (int k = kb; k < ke; ++k) slots[4 * (size_t)(-2 - emark[sarr[k]]) + slid[k]] = (short)(k - kb);
        for (size_t q = 0; q < elems.size(); ++q) {
            const int e = elems[q];
            emark[e] = b;
            int nmin = nn;
            unsigned short ln[4];
            for (int i = 0; i < 4; ++i) {
                const int n = conn[(size_t)i*ne + e];
                nmin = std::min(nmin, n);
                if (n >= n0 && n < n1) ln[i] = (unsigned short)(n - n0);
                else ln[i] = (unsigned short)(nown + (std::lower_bound(halo.begin(), halo.end(), n) - halo.begin()));
            }
            P.pe_elem.push_back(e | ((nmin >= n0 && nmin < n1) ? 0x40000000 : 0));
            {
                ulonglong2 r;
                r.x = (unsigned long long)(unsigned)P.pe_elem.back() | ((unsigned long long)ln[0] << 31) | ((unsigned long long)ln[1] << 40)
                      | ((unsigned long long)ln[2] << 49);
                r.y = (unsigned long long)ln[3];
                for (int k = 0; k < 4; ++k)
                    r.y |= (unsigned long long)(slots[4*q + k] < 0 ? 0xfff : slots[4*q + k]) << (9 + 12 * k);
                P.pe_pack.push_back(r);
            }
        }
        P.pn_id.insert(P.pn_id.end(), halo.begin(), halo.end());
        P.pe_ptr.push_back((int)P.pe_elem.size());
        P.pn_ptr.push_back((int)P.pn_id.size());
        P.max_inc = std::max(P.max_inc, ke - kb);
        P.max_pn = std::max(P.max_pn, nown + (int)halo.size());
        P.max_pe = std::max(P.max_pe, (int)elems.size());
    }
    return true;
}
