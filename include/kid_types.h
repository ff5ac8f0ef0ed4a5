/* kid_types.h -- plain-data interface types of the KID (Kinematic Iceberg Dynamics) evolve-loop boundary.
 *
 * These are interface types only (no algorithm).  They describe, as flat C structs, the pieces of the
 * reference's Fortran derived types that the per-berg evolve loop reads and writes:
 *   - type icebergs_gridded   /root/reference/src/icebergs_framework.F90:112-229  -> kid_grid_desc + field enums
 *   - type iceberg            /root/reference/src/icebergs_framework.F90:290-359  -> kid_berg_soa (structure of arrays)
 *   - scalar members of type icebergs and the module switches
 *                             /root/reference/src/icebergs_framework.F90:28-64, 421-616 -> kid_params
 *
 * All arrays are column-major exactly as Fortran owns them: a gridded field f(isd:ied,jsd:jed) is
 * passed as a pointer to f(isd,jsd); element (i,j) lives at [(i-isd) + (j-jsd)*(ied-isd+1)].
 * "real" in the reference is fp64 (the FMS build promotes with -r8), integers are int32 except berg ids.
 */
#ifndef KID_TYPES_H
#define KID_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- grid descriptor: index ranges + geometry switches (FW:112-140, FW:748-749, FW:712) ---- */
typedef struct kid_grid_desc {
  int32_t isd, ied, jsd, jed;   /* data domain (with halo)   */
  int32_t isc, iec, jsc, jec;   /* computational domain      */
  int32_t grid_is_latlon;       /* FW:748 */
  int32_t grid_is_regular;      /* FW:749 */
  /* Where this grid sits in the global one (a domain-decomposed host: one handle per tile): gni = global zonal size (icebergs_init's
   * gni, IB:92-99), gi0 / gj0 = what must be added to a local cell index to get the global one.  They enter the cell hash of a
   * berg id only (ij_component_of_id FW:4227-4240: i_global + gni * (j_global - 1)), so that bergs created on different tiles
   * (calving, footloose children, ids of a 32-bit-era restart) never share an id.  gni = 0: this grid is the whole grid. */
  int32_t gni, gnj, gi0, gj0;
  double  Lx;                   /* FW:712 zonal period; <=0 means not periodic */
} kid_grid_desc;

/* ---- static grid fields, grd%* after ice_bergs_framework_init (FW:1021-1094) ---- */
enum {
  KID_G_LON = 0,   /* grd%lon  : NE-corner longitude of cell (i,j)  */
  KID_G_LAT,       /* grd%lat  */
  KID_G_LONC,      /* grd%lonc : cell-centre longitude              */
  KID_G_LATC,
  KID_G_DX, KID_G_DY, KID_G_AREA, KID_G_MSK, KID_G_COS, KID_G_SIN, KID_G_OCEAN_DEPTH,
  KID_NGRID_STATIC
};

/* ---- forcing fields, grd%* after the ingest block of icebergs_run (IB:5236-5383) ---- */
enum {
  KID_F_UO = 0, KID_F_VO, KID_F_UI, KID_F_VI, KID_F_UA, KID_F_VA,
  KID_F_SSH, KID_F_SST, KID_F_SSS, KID_F_CN, KID_F_HI,
  KID_NFORCING
};

/* ---- forcing ingest: the arguments of icebergs_run as the coupler passes them (IB:5074-5096), turned
 * into the grd%* planes above by kid_ingest_forcing (IB:5236-5383).  Arrays are column-major, first index fastest.
 * Extents: uo, ui are u_ni x u_nj and vo, vi are v_ni x v_nj (B-grid: both (isc-1:iec+1, jsc-1:jec+1); C-grid:
 * symmetric memory or not, the offsets follow IB:5246-5249); tauxa is taux_ni x taux_nj, tauya tauy_ni x tauy_nj;
 * ssh, cn, hi cover (isc-1:iec+1, jsc-1:jec+1); sst, sss cover the computational domain; sss may be NULL
 * (grd%sss = -1, IB:5357). ---- */
enum { KID_BGRID_NE = 0, KID_CGRID_NE = 1, KID_AGRID = 2 };   /* mpp_parameter_mod staggers the reference accepts */
typedef struct kid_forcing_in {
  const double *uo, *vo, *ui, *vi;
  const double *tauxa, *tauya;
  const double *ssh, *sst, *cn, *hi, *sss;
  int32_t u_ni, u_nj, v_ni, v_nj;           /* size(uo,1), size(uo,2), size(vo,1), size(vo,2) */
  int32_t taux_ni, taux_nj, tauy_ni, tauy_nj;
  int32_t vel_stagger, stress_stagger;      /* stagger / stress_stagger arguments, IB:5090-5091 */
  int32_t tau_is_velocity;                  /* bergs%tau_is_velocity, IB:5321 */
  int32_t cyclic_x;                         /* the domain is zonally cyclic: halo columns wrap (mpp_update_domains on one rank) */
  int32_t on_device, pad;                   /* pointers are device addresses (no staging copy) */
} kid_forcing_in;

/* ---- calving source: the calving block of icebergs_run IB:5203-5231, accumulate_calving IB:6153-6222 and
 * calve_icebergs IB:6225-6402.  kid_calving_params holds the per-hemisphere namelist tables as
 * ice_bergs_framework_init leaves them (FW:1534-1551; initial_mass_s/n are in kid_params). ---- */
enum { KID_NCLASSES = 10 };     /* nclasses, FW:36 */
typedef struct kid_calving_params {
  double distribution_s[10], distribution_n[10];            /* FW:1535, 1544 */
  double mass_scaling_s[10], mass_scaling_n[10];            /* FW:1536, 1545 */
  double initial_thickness_s[10], initial_thickness_n[10];  /* FW:1537, 1546 */
  double initial_width_s[10], initial_width_n[10];          /* FW:1540, 1549 */
  double initial_length_s[10], initial_length_n[10];        /* FW:1541, 1550 */
  double tau_calving;           /* bergs%tau_calving, IB:5215 (>0: running-mean calving, IB:5999-6038) */
  int32_t restarted;            /* bergs%restarted: skips the first-call stored-heat initialisation, IB:6171 */
  int32_t pad;
} kid_calving_params;
typedef struct kid_calving_in {
  const double *calving;        /* (isc:iec, jsc:jec), kg/m2/s, IB:5078 */
  const double *calving_hflx;   /* (isc:iec, jsc:jec), W/m2,    IB:5080 */
  int32_t on_device, pad;       /* pointers are device addresses */
} kid_calving_in;
/* ---- trajectory sampling: which bergs record_posn samples and what iceberg_trajectories.nc holds
 * (FW:5328-5498, icebergs_fms2io.F90:1631-2103); namelist values ---- */
typedef struct kid_traj_params {
  double traj_area_thres;                        /* km^2, FW:687 */
  double traj_area_thres_sntbc;                  /* km^2, FW:688 */
  double traj_area_thres_fl;                     /* km^2, FW:689 */
  double save_all_traj_year;                     /* FW:763 */
  double save_traj_by_class_start_mass_thres_s;  /* kg, FW:5375 */
  double save_traj_by_class_start_mass_thres_n;  /* kg, FW:5377 */
  int32_t save_short_traj;                       /* FW:759 */
  int32_t save_fl_traj;                          /* FW:762 */
  int32_t save_nonfl_traj_by_class;              /* FW:5371 */
  int32_t save_bond_traj;                        /* FW:49: also sample every bond of a sampled berg */
} kid_traj_params;

/* what one kid_calving call adds to the budget scalars of type icebergs (increments; the first two `stored` entries
 * are the values the first call prints, zero afterwards) */
enum {
  KID_CS_NET_CALVING_RECEIVED = 0,        /* IB:5204 */
  KID_CS_NET_INCOMING_CALVING,            /* IB:5223 */
  KID_CS_NET_INCOMING_CALVING_HEAT,       /* IB:5231 */
  KID_CS_STORED_START,                    /* IB:6176 */
  KID_CS_STORED_HEAT_START,               /* IB:6186 */
  KID_CS_NET_CALVING_USED,                /* IB:6212 */
  KID_CS_NET_INCOMING_CALVING_HEAT_USED,  /* IB:6218 */
  KID_CS_UNUSED_CALVING,                  /* IB:5397 (a value, not an increment) */
  KID_CS_NET_CALVING_TO_BERGS,            /* IB:6399 */
  KID_CS_NET_HEAT_TO_BERGS,               /* IB:6400 */
  KID_CS_NBERGS_CALVED,                   /* IB:6383 */
  KID_CS_ERROR_COUNT,                     /* FATALs of the block (berg not in its cell IB:6281) + SoA overflow */
  KID_CS_NBERGS_CALVED_BY_CLASS_S,        /* 10 entries, IB:6385 */
  KID_CS_NBERGS_CALVED_BY_CLASS_N = KID_CS_NBERGS_CALVED_BY_CLASS_S + 10,   /* 10 entries, IB:6387 */
  KID_NCALV_SCALARS = KID_CS_NBERGS_CALVED_BY_CLASS_N + 10
};

/* ---- per-berg fp64 fields (type iceberg, FW:294-343) ---- */
enum {
  KID_B_LON = 0, KID_B_LAT, KID_B_UVEL, KID_B_VVEL,
  KID_B_MASS, KID_B_THICKNESS, KID_B_WIDTH, KID_B_LENGTH,
  KID_B_AXN, KID_B_AYN, KID_B_BXN, KID_B_BYN,
  KID_B_UVEL_PREV, KID_B_VVEL_PREV, KID_B_UVEL_OLD, KID_B_VVEL_OLD, KID_B_LON_OLD, KID_B_LAT_OLD,
  KID_B_START_LON, KID_B_START_LAT, KID_B_START_DAY, KID_B_START_MASS,
  KID_B_MASS_SCALING, KID_B_MASS_OF_BITS, KID_B_MASS_OF_FL_BITS, KID_B_MASS_OF_FL_BERGY_BITS,
  KID_B_FL_K, KID_B_HEAT_DENSITY, KID_B_HALO_BERG, KID_B_STATIC_BERG,
  KID_B_XI, KID_B_YJ,
  /* environment as seen by the berg (FW:331-343) */
  KID_B_UO, KID_B_VO, KID_B_UI, KID_B_VI, KID_B_UA, KID_B_VA,
  KID_B_SSH_X, KID_B_SSH_Y, KID_B_SST, KID_B_SSS, KID_B_CN, KID_B_HI, KID_B_OD,
  /* multiple-time-stepping / DEM extras (FW:348-358) */
  KID_B_AXN_FAST, KID_B_AYN_FAST, KID_B_BXN_FAST, KID_B_BYN_FAST, KID_B_ANG_VEL, KID_B_ANG_ACCEL, KID_B_ROT,
  KID_NB_F64
};
/* directions of send_bergs_to_other_pes (FW:3022-3130): east and west are exchanged first, then north and south */
enum { KID_DIR_E = 0, KID_DIR_W, KID_DIR_N, KID_DIR_S };
/* ---- per-berg int32 fields ---- */
enum { KID_BI_INE = 0, KID_BI_JNE, KID_BI_START_YEAR, KID_BI_N_BONDS, KID_BI_ALIVE, KID_BI_CONGLOM_ID, KID_NB_I32 };

/* Structure of arrays: one contiguous array of length n per field.  A NULL pointer on upload means
 * "all zero"; on download it means "do not copy this field back". */
typedef struct kid_berg_soa {
  int64_t  n;
  double  *f64[KID_NB_F64];
  int32_t *i32[KID_NB_I32];
  int64_t *id;               /* FW:325 integer(kind=8) :: id */
} kid_berg_soa;

/* ---- bonds (type bond, FW:362-386): the per-berg linked list first_bond -> next_bond flattened to `count[k]` slots,
 * in list order.  Arrays are slot-major: x[s * n + k] is slot s of berg k, s < max_bonds.  A bond exists on both of
 * its bergs (the reference's matching `other_bond`); the library re-derives row indices from other_id. ---- */
enum {
  KID_BOND_LENGTH = 0, KID_BOND_TANGD1, KID_BOND_TANGD2, KID_BOND_NSTRESS, KID_BOND_SSTRESS, KID_BOND_REL_ROTATION,
  KID_BOND_F_X, KID_BOND_F_Y, KID_BOND_FD_X, KID_BOND_FD_Y, KID_BOND_T, KID_BOND_T_D,   /* save_bond_forces, FW:379-385 */
  KID_NBOND_F64
};
enum { KID_MAX_BONDS = 6 };                /* max_bonds FW:693 */
typedef struct kid_bond_soa {
  int64_t  n;                              /* bergs (rows) */
  int32_t  max_bonds; int32_t pad;         /* slots per berg, <= KID_MAX_BONDS */
  int32_t *count;                          /* [n] bonds in berg k's list, broken ones included */
  int64_t *other_id;                       /* [max_bonds*n] FW:367 */
  int32_t *broken;                         /* [max_bonds*n] FW:377 */
  double  *f64[KID_NBOND_F64];             /* [max_bonds*n] each; NULL = zero on upload / skip on download */
} kid_bond_soa;

/* ---- per-cell accumulators written by the hot path (zeroed at IB:5125-5156) ----
 * Planes that are always live come first (KID_NACC_CORE of them): that prefix is what gets zeroed each step
 * and what the multi-GPU all-reduce carries; the diagnostics-only planes (guarded by `id_*>0` in the
 * reference) follow and join the reduction only when kid_params.diag_mask asks for them. */
enum {
  KID_A_FLOATING_MELT = 0, /* IB:3117 */
  KID_A_BERG_MELT,         /* IB:3133 */
  KID_A_CALVING_HFLX,      /* IB:3129 (increment only; the caller adds the masked coupler input, IB:5207) */
  KID_A_BERGY_SRC,         /* IB:3136 */
  KID_A_BERGY_MELT,        /* IB:3139 */
  KID_A_FL_BITS_MELT,      /* IB:3142 */
  KID_A_FL_BITS_SRC,       /* IB:3287, IB:2642 */
  KID_A_BERGY_MASS, KID_A_FL_BITS_MASS, KID_A_FL_BERGY_BITS_MASS, /* IB:5060-5070 (on with add_weight_to_ocean) */
  KID_A_MASS_ON_OCEAN,                       /* 9 consecutive slots, IB:4088 */
  KID_A_AREA_ON_OCEAN = KID_A_MASS_ON_OCEAN + 9,
  KID_A_UVEL_ON_OCEAN = KID_A_AREA_ON_OCEAN + 9,
  KID_A_VVEL_ON_OCEAN = KID_A_UVEL_ON_OCEAN + 9,
  KID_NACC_CORE = KID_A_VVEL_ON_OCEAN + 9,
  /* diagnostics-only planes */
  KID_A_MASS = KID_NACC_CORE, KID_A_VIRTUAL_AREA, KID_A_U_ICEBERG, KID_A_V_ICEBERG, /* IB:5026-5058 */
  KID_A_MELT_BUOY, KID_A_MELT_EROS, KID_A_MELT_CONV,             /* IB:3154-3165 */
  KID_A_MELT_BUOY_FL, KID_A_MELT_EROS_FL, KID_A_MELT_CONV_FL,    /* IB:3167-3198 */
  KID_A_FL_PARENT_MELT, KID_A_FL_CHILD_MELT,                     /* IB:3146-3153, 3183-3186 */
  KID_A_MELT_BY_CLASS,                       /* 10 classes, IB:3125 */
  KID_NACC = KID_A_MELT_BY_CLASS + 10
};
/* ---- derived gridded outputs of create_gridded_icebergs_fields (IB:3390-3489) ---- */
enum {
  KID_O_SPREAD_MASS = 0, KID_O_SPREAD_AREA, KID_O_SPREAD_UVEL, KID_O_SPREAD_VVEL, KID_O_USTAR_ICEBERG,
  KID_NOUT
};
/* ---- scalars mutated on `bergs` by the hot path ---- */
enum {
  KID_S_NET_HEAT_TO_OCEAN = 0,  /* IB:3130 */
  KID_S_NBERGS_MELTED,          /* IB:3295 */
  KID_S_NBERGS_CALVED_FL,       /* IB:2634, IB:3275 */
  KID_S_NSPEEDING_TICKETS,      /* IB:2314 */
  KID_S_NBERGS_ALIVE,           /* bookkeeping of the SoA (not in the reference) */
  KID_S_ERROR_COUNT,            /* bergs that hit a reference FATAL/WARNING path (e.g. IB:3207, FW:6502) */
  KID_S_NBONDS_BROKEN,          /* bond sides that fractured (bond_break_detected IB:1144, counted) */
  KID_NSCALAR = 8
};

/* diagnostics that the reference guards with `id_*>0` (diag_manager ids, FW:1567-1673) */
enum {
  KID_DIAG_MELT_BY_CLASS  = 1 << 0,  /* IB:3119 */
  KID_DIAG_FL_PARENT_MELT = 1 << 1,  /* IB:3146 */
  KID_DIAG_FL_CHILD_MELT  = 1 << 2,
  KID_DIAG_MELT_BUOY      = 1 << 3,
  KID_DIAG_MELT_EROS      = 1 << 4,
  KID_DIAG_MELT_CONV      = 1 << 5,
  KID_DIAG_MELT_BUOY_FL   = 1 << 6,
  KID_DIAG_MELT_EROS_FL   = 1 << 7,
  KID_DIAG_MELT_CONV_FL   = 1 << 8,
  KID_DIAG_VIRTUAL_AREA   = 1 << 9,  /* IB:5026 */
  KID_DIAG_MASS           = 1 << 10, /* IB:5051 */
  KID_DIAG_U_ICEBERG      = 1 << 11,
  KID_DIAG_V_ICEBERG      = 1 << 12,
  KID_DIAG_BERGY_MASS     = 1 << 13, /* IB:5061 (also on when add_weight_to_ocean) */
  KID_DIAG_FL_BITS_MASS   = 1 << 14,
  KID_DIAG_FL_BERGY_BITS_MASS = 1 << 15,
  KID_DIAG_SPREAD_UVEL    = 1 << 16, /* IB:3419 */
  KID_DIAG_SPREAD_VVEL    = 1 << 17,
  KID_DIAG_SPREAD_AREA    = 1 << 18,
  KID_DIAG_USTAR_ICEBERG  = 1 << 19  /* IB:3466 */
};

enum { KID_FL_STYLE_NEW_BERGS = 0, KID_FL_STYLE_FL_BITS = 1 };

/* ---- scalar parameters: namelist icebergs_nml (FW:825-856, defaults FW:686-822) + FMS constants ---- */
typedef struct kid_params {
  /* FMS constants_mod values (not in the reference tree; pinned) */
  double pi, omega, HLF;
  /* time */
  double dt;                      /* bergs%dt */
  int32_t current_year; int32_t pad0;
  double current_yearday;
  /* physics scalars */
  double Rearth;                  /* FW:44  */
  double rho_bergs;               /* FW:694 */
  double lat_ref;                 /* FW:709 */
  double cdrag_grounding;         /* FW:697 */
  double h_to_init_grounding;     /* FW:698 */
  double ocean_drag_scale;        /* FW:813 */
  double speed_limit;             /* FW:726 */
  double sicn_shift;              /* FW:708 */
  double bergy_bit_erosion_fraction; /* FW:707 */
  double tip_parameter;           /* FW:729 */
  double grounding_fraction;      /* FW:730 */
  double clipping_depth;          /* FW:227 */
  double coastal_drift;           /* FW:731 */
  double tidal_drift;             /* FW:732 (must be 0: the stochastic stream is FMS's) */
  double initial_orientation;     /* FW:713 */
  double melt_cutoff;             /* FW:718 */
  double cdrag_icebergs;          /* FW:716 */
  double utide_icebergs;          /* FW:714 */
  double ustar_icebergs_bg;       /* FW:715 */
  double Gamma_T_3EQ;             /* FW:717 */
  double fl_youngs;               /* FW:817 */
  double fl_strength;             /* FW:818 */
  double new_berg_from_fl_bits_mass_thres; /* FW:822 */
  double u_override, v_override;  /* FW:710-711 */
  double initial_mass_s[10];      /* FW:787 */
  double initial_mass_n[10];      /* FW:793 */
  /* interactions, multiple time stepping, DEM (values as left by ice_bergs_framework_init) */
  double spring_coef;             /* FW:695 */
  double contact_spring_coef;     /* FW:696 (= spring_coef when the namelist leaves it 0) */
  double contact_distance;        /* FW:782 */
  double radial_damping_coef;     /* FW:704 */
  double tangental_damping_coef;  /* FW:705 */
  double convergence_tolerance;   /* FW:785 */
  double constant_length, constant_width; /* FW:811-812 */
  double dem_spring_coef;         /* FW:804 */
  double dem_damping_coef;        /* FW:805 */
  double poisson;                 /* FW:803 */
  double dem_tests_start_lon;     /* FW:595, 4707: start_lon of the west-most element (dem_tests_init; the host sets it, with every berg's start_lon = lon) */
  double dem_tests_end_lon;       /* FW:596, 4707: ... of the east-most one */
  double frac_thres_n, frac_thres_t; /* FW:1355-1356 (already scaled by frac_thres_scaling) */
  /* switches (0/1) */
  int32_t Runge_not_Verlet;             /* FW:733 */
  int32_t use_new_predictive_corrective;/* FW:770 */
  int32_t old_interp_flds_order;        /* FW:1483 (derived) */
  int32_t old_bug_bilin;                /* FW:40  */
  int32_t use_f_plane;                  /* FW:747 */
  int32_t use_operator_splitting;       /* FW:720 */
  int32_t add_weight_to_ocean;          /* FW:721 */
  int32_t time_average_weight;          /* FW:723 */
  int32_t use_old_spreading;            /* FW:778 */
  int32_t hexagonal_icebergs;           /* FW:753 */
  int32_t allow_bergs_to_roll;          /* FW:752 */
  int32_t use_updated_rolling_scheme;   /* FW:738 */
  int32_t set_melt_rates_to_zero;       /* FW:751 */
  int32_t use_mixed_melting;            /* FW:734 */
  int32_t melt_icebergs_as_ice_shelf;   /* FW:743 */
  int32_t Use_three_equation_model;     /* FW:742 */
  int32_t use_mixed_layer_salinity_for_thermo; /* FW:740 */
  int32_t const_gamma;                  /* FW:719 */
  int32_t apply_thickness_cutoff_to_bergs_melt;   /* FW:737 */
  int32_t apply_thickness_cutoff_to_gridded_melt; /* FW:736 */
  int32_t Iceberg_melt_without_decay;   /* FW:744 */
  int32_t find_melt_using_spread_mass;  /* FW:741 */
  int32_t override_iceberg_velocities;  /* FW:746 */
  int32_t iceberg_bonds_on;             /* FW:51  */
  int32_t internal_bergs_for_drag;      /* FW:735 */
  int32_t dem;                          /* FW:52  */
  int32_t mts;                          /* FW:48  */
  int32_t footloose;                    /* FW:64  */
  int32_t fl_style;                     /* FW:820 (KID_FL_STYLE_*) */
  int32_t fl_bits_erosion_to_bergy_bits;/* FW:821 */
  int32_t displace_fl_bergs;            /* FW:819 */
  int32_t use_roundoff_fix;             /* FW:37  */
  int32_t interactive_icebergs_on;      /* FW:771 */
  int32_t only_interactive_forces;      /* FW:757 */
  int32_t pass_fields_to_ocean_model;   /* FW:739 */
  int32_t static_icebergs;              /* FW:756 */
  int32_t old_bug_rotated_weights;      /* FW:38  */
  int32_t mts_sub_steps;                /* FW:780, 1296-1301 (>0; mts_fast_dt = dt/mts_sub_steps) */
  int32_t explicit_inner_mts;           /* FW:784 (forced on by dem, FW:1433) */
  int32_t force_convergence;            /* FW:783 */
  int32_t critical_interaction_damping_on; /* FW:773 */
  int32_t tang_crit_int_damp_on;        /* FW:774 */
  int32_t scale_damping_by_pmag;        /* FW:772 */
  int32_t contact_cells_lon, contact_cells_lat; /* FW:1514-1518 */
  int32_t constant_interaction_LW;      /* FW:810 */
  int32_t ignore_tangential_force;      /* FW:802 */
  int32_t fracture_criterion_stress;    /* FW:800: 1 = 'stress', 0 = 'none' */
  int32_t max_bonds;                    /* FW:693 */
  int32_t use_broken_bonds_for_substep_contact; /* FW:806, 1439-1447 */
  int32_t break_bonds_on_sub_steps;     /* FW:61 */
  int32_t short_step_mts_grounding;     /* FW:54 */
  int32_t use_grounding_torque;         /* FW:801 */
  int32_t radius_based_drag;            /* FW:55 */
  int32_t orig_dem_moment_of_inertia;   /* FW:60 */
  int32_t rev_mind;                     /* FW:59 */
  int32_t rotate_icebergs_for_mass_spreading; /* FW:750: hexagon orientation from the bonds (IB:4004) */
  int32_t diag_mask;                    /* KID_DIAG_* : which `id_*>0` guards are on */
  int32_t periodic_reentry;             /* the handle owns the whole zonal period (Lx > 0): a berg that leaves through the east or
                                           west edge of the computational domain re-enters on the other side the way it does where the
                                           seam is a boundary between two PEs (send_bergs_to_other_pes FW:3024-3041 ->
                                           unpack_berg_from_buffer2 FW:3573-3577, 3628-3635), and the 9-point gather reads across the
                                           seam (mpp_update_domains in sum_up_spread_fields, IB:6103).  0: the berg is removed, as
                                           on a PE without that neighbour */
  int32_t fl_init_child_xy_by_pe;       /* FW:606, 816: one random number for the whole run (old bug) instead of one per calving event */
  int32_t fl_rng_seed;                  /* seed of the counter-based generator that places footloose children (include/kid_rng.h); the
                                           reference seeds FMS's stream from (mpp_pe(), time), IB:2548 */
  int32_t dem_beam_test;                /* FW:808: 1 = simply supported beam, 2 = cantilever beam (the loads of IB:1861-1877); 0 = off.
                                           (An even number of int32 members: no implicit tail padding, Fortran stream I/O moves components) */
} kid_params;

#ifdef __cplusplus
}
#endif
#endif /* KID_TYPES_H */
